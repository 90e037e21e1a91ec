"""Oracle: PixelNeRF baseline decoder render (vanilla_nerf/model_pixel.py:133-258): pixel-aligned
latents of NV source views, late-fusion MLP, vanilla coarse+fine sampling and compositing.
Test infrastructure (oracle/__init__.py).  The image encoder is outside the hot path; its
output latent (NV,512,Hf,Wf) is an input here (`scene`)."""
import torch

from . import compositing, encoding, gather, mlp, sampling


def region_eval(params, prefix, rays, scene, tvals, sigma_noise=None):
    """Per-point outputs at GIVEN sample positions t (B,N): rgb (B,N,3) = sigmoid(raw), sigma (B,N,1) = relu(raw).
    model_pixel.py:198-237: points o + t*rays_d; latent lookup with view 0's focal (f, f) / centre for
    all views; pos_enc of the CAMERA-frame point; view directions in the camera frame, tiled with
    (1,N,1) so row (view, ray b, sample s) carries the direction of ray (b*N+s) mod B."""
    o, d, vd = rays["rays_o"], rays["rays_d"], rays["viewdirs"]
    poses, focal, centre = rays["src_poses"], rays["src_focal"], rays["src_c"]
    nv = poses.shape[0]
    B, N = tvals.shape
    pts = sampling.points_on_rays(tvals, o, d)
    local = gather.pixel_aligned_features(pts, scene["latent"], poses, focal, centre, scene["image_wh"], flip_y=False)
    cam = gather.world_to_camera(pts.reshape(-1, 3), poses)
    x_enc = encoding.pos_enc(cam, 0, 10)
    d_enc = encoding.pos_enc(gather.world_to_camera_dirs(vd, poses), 0, 4)
    d_rows = torch.tile(d_enc[:, None, :], (1, N, 1)).reshape(-1, d_enc.shape[-1])
    raw_rgb, raw_sigma = mlp.pixelnerf_mlp(params, prefix, x_enc, d_rows, local, nv)
    if sigma_noise is not None:
        raw_sigma = raw_sigma.reshape(B, N, -1) + sigma_noise.reshape(B, N, 1)
    return torch.sigmoid(raw_rgb.reshape(B, N, -1)), torch.relu(raw_sigma.reshape(B, N, -1))


def render(params, rays, scene, near, far, n_coarse=64, n_fine=64, white_bkgd=False, keep=False, samples=None, sigma_noise=None):
    """[(rgb (B,3), acc (B,), depth (B,))] x 2 = PixelNeRF.forward for randomized=False.
    samples = (t0 (B,n_coarse+1), t1 (B,n_coarse+1+n_fine)): evaluate at GIVEN sample positions instead of the deterministic
    ones (the randomized call's draws, taken from the implementation under test; they carry no gradient, helper.py:610-616).
    sigma_noise = per level (B,N) values added to the raw density (model_pixel.py:235-236: rand_like * noise_std)."""
    o, d = rays["rays_o"], rays["rays_d"]
    out, extra = [], []
    t = w = None
    for level, prefix in enumerate(("coarse_mlp.", "fine_mlp.")):
        if samples is not None:
            t = samples[level]
        elif level == 0:
            t, _ = sampling.vanilla_level0(o, d, n_coarse, near, far)
        else:
            mids = 0.5 * (t[..., 1:] + t[..., :-1])
            t, _ = sampling.vanilla_level1(mids, w[..., 1:-1], o, d, t, n_fine)
        rgb, sigma = region_eval(params, prefix, rays, scene, t, None if sigma_noise is None else sigma_noise[level])
        comp, acc, w, depth = compositing.vanilla_composite(rgb, sigma, t, d, white_bkgd)
        out.append((comp, acc, depth))
        extra.append(dict(t=t, sigma=sigma, rgb=rgb, weights=w))
    return (out, extra) if keep else out


_WHOLE = ("src_imgs", "src_poses", "src_focal", "src_c")


def render_chunked(params, rays, scene, near, far, chunk, **kw):
    B = rays["rays_o"].shape[0]
    rgb, depth = [], []
    for i in range(0, B, chunk):
        part = {k: (v if k in _WHOLE else v[i:i + chunk]) for k, v in rays.items()}
        res = render(params, part, scene, near, far, **kw)
        rgb.append(res[1][0])
        depth.append(res[1][2])
    return torch.cat(rgb, 0), torch.cat(depth, 0)
