"""Oracle: NeO-360 decoder render (tri-planar + pixel-aligned features,
inside/outside-sphere two-region NeRF++).  Test infrastructure (oracle/__init__.py).

The scene encoder (GridEncoder / ResNet) is outside the hot path; its outputs —
three tri-planes (NV,128,Hp,Wp) and the pixel-aligned latent (NV,512,Hf,Wf) —
are inputs here (`scene`).
"""
import torch

from . import compositing, encoding, gather, mlp, rays as rays_mod, sampling


def _predict(params, prefix, cam_pts, dir_cam, world, local, B, N, nv):
    """neo360/model.py:343-407.  Reproduces the view-direction tiling of
    :357-360 literally: the (NV,B,27) encoding is tiled with (1,N,1), so the
    row for (view v, ray b, sample s) carries the direction of ray
    (b*N+s) mod B of THIS call."""
    x_enc = encoding.pos_enc(cam_pts, 0, 10)
    d_enc = encoding.pos_enc(dir_cam, 0, 4)
    d_rows = torch.tile(d_enc[:, None, :], (1, N, 1)).reshape(-1, d_enc.shape[-1])
    raw_rgb, raw_sigma = mlp.nerfpp_mlp(params, prefix, x_enc, d_rows, world, local, nv)
    return mlp.colour_activation(raw_rgb.reshape(B, N, -1)), mlp.density_activation(raw_sigma.reshape(B, N, -1))


def region_eval(params, prefix, rays, scene, tvals, inside, far=None):
    """One region's per-point outputs at GIVEN sample positions: rgb (B,N,3), sigma (B,N,1).
    inside: tvals = t ascending, points o + t d.  outside: tvals = inverse radius
    descending; encoded point = inverted-sphere point, lookups at the linear point
    o + (far(1-s) + 3 s) d (neo360/helper.py:59-73, model.py:409-464).
    Used by the stage-isolated parity tests (identical sample positions on both sides)."""
    o, d, vd = rays["rays_o"], rays["rays_d"], rays["viewdirs"]
    poses, focal, centre = rays["src_poses"], rays["src_focal"], rays["src_c"]
    nv = poses.shape[0]
    dir_cam = gather.world_to_camera_dirs(vd, poses)
    B, N = tvals.shape
    if inside:
        look = enc = sampling.points_on_rays(tvals, o, d)
        extra = None
    else:
        p4 = sampling.inverted_sphere_points(o, d, tvals)
        enc = p4[:, :, :3]
        extra = p4[:, :, 3].reshape(-1, 1).unsqueeze(0).repeat(nv, 1, 1)
        look = sampling.points_on_rays(far * (1.0 - tvals) + 3.0 * tvals, o, d)
    world = gather.triplane_features(look, scene["plane_xz"], scene["plane_xy"], scene["plane_yz"], poses)
    local = gather.pixel_aligned_features(look, scene["latent"], poses, focal, centre, scene["image_wh"])
    cam = gather.world_to_camera(enc.reshape(-1, 3), poses)
    if extra is not None:
        cam = torch.cat((cam, extra), dim=-1)
    return _predict(params, prefix, cam, dir_cam, world, local, B, N, nv)


def render(params, rays, scene, n_coarse=128, n_fine=256, white_bkgd=False, out_depth=True, keep=False, uniforms=None,
           fine_samples=None):
    """Return value of NeRF_TP.forward for randomized=False
    (neo360/model.py:266-581, decoder half from :276).

    rays: rays_o, rays_d, viewdirs (B,3), src_poses (NV,4,4), src_focal (NV,),
    src_c (NV,2).  scene: plane_xz/xy/yz (NV,C,Hp,Wp), latent (NV,512,Hf,Wf),
    image_wh (W,H).  Per level: out_depth -> (rgb, fg_rgb, bg_rgb, fg_acc,
    bg_lambda, depth); else (rgb, fg_w, bg_w, fg_sdist, bg_sdist, bg_acc).
    uniforms: None = randomized=False; else dict(fg0, bg0 (B,n_coarse+1), fg1, bg1 (B,n_fine)) = the draws the
    reference's randomized=True path takes from torch.rand, in call order (helper.py:49 twice, :196 twice).
    fine_samples: None, or (fg_t, bg_s) (B, n_coarse+1+n_fine) = level-1 sample positions to USE instead of resampling
    (derivative checks: the resampled positions carry no gradient - helper.py:224 detaches them - but their value decides
    which texels a sample blends, so a gradient comparison between two arithmetics wants both sides at the same positions).
    """
    from . import training
    o, d, vd = rays["rays_o"], rays["rays_d"], rays["viewdirs"]
    poses, focal, centre = rays["src_poses"], rays["src_focal"], rays["src_c"]
    nv = poses.shape[0]
    near = torch.full_like(o[..., -1:], 1e-4)                      # model.py:277
    far, _ = rays_mod.sphere_exit_depth(o, d)                      # model.py:278
    dir_cam = gather.world_to_camera_dirs(vd, poses)               # model.py:339-341
    planes = (scene["plane_xz"], scene["plane_xy"], scene["plane_yz"])

    def feats(p3):
        return (gather.triplane_features(p3, *planes, poses),
                gather.pixel_aligned_features(p3, scene["latent"], poses, focal, centre, scene["image_wh"]))

    out, extra = [], []
    fg_t = bg_s = fg_w = bg_w = None
    for level in range(2):
        if level == 0 and uniforms is None:
            fg_t, fg_p = sampling.neo_fg_level0(o, d, n_coarse, near, far)
            bg_s, bg_p4, bg_lin = sampling.neo_bg_level0(o, d, n_coarse, far, 3.0)
            fg_name, bg_name = "fg_coarse_mlp.", "bg_coarse_mlp."
        elif level == 1 and uniforms is None and fine_samples is not None:
            fg_t, bg_s = fine_samples
            fg_p = sampling.points_on_rays(fg_t, o, d)
            bg_p4 = sampling.inverted_sphere_points(o, d, bg_s)
            bg_lin = sampling.points_on_rays(far * (1.0 - bg_s) + 3.0 * bg_s, o, d)
            fg_name, bg_name = "fg_fine_mlp.", "bg_fine_mlp."
        elif level == 1 and uniforms is None:
            fg_mid = 0.5 * (fg_t[..., 1:] + fg_t[..., :-1])
            bg_mid = 0.5 * (bg_s[..., 1:] + bg_s[..., :-1])
            fg_t, fg_p = sampling.neo_fg_level1(fg_mid, fg_w[..., 1:-1], o, d, fg_t, n_fine)
            bg_s, bg_p4, bg_lin = sampling.neo_bg_level1(bg_mid, bg_w[..., 1:-1], o, d, bg_s, n_fine, far, 3.0)
            fg_name, bg_name = "fg_fine_mlp.", "bg_fine_mlp."
        else:
            # randomized=True: sample rows from the given draws, points exactly as the deterministic branches build them
            if level == 0:
                fg_t, bg_s = training.neo_level0_randomized(far, n_coarse, uniforms["fg0"], uniforms["bg0"])
                fg_name, bg_name = "fg_coarse_mlp.", "bg_coarse_mlp."
            else:
                fg_t = training.resample_randomized(fg_t, fg_w, uniforms["fg1"], descending=False)
                bg_s = training.resample_randomized(bg_s, bg_w, uniforms["bg1"], descending=True)
                fg_name, bg_name = "fg_fine_mlp.", "bg_fine_mlp."
            fg_p = sampling.points_on_rays(fg_t, o, d)
            bg_p4 = sampling.inverted_sphere_points(o, d, bg_s)
            bg_lin = sampling.points_on_rays(far * (1.0 - bg_s) + 3.0 * bg_s, o, d)
        B, N, _ = fg_p.shape
        fg_world, fg_local = feats(fg_p)
        bg_world, bg_local = feats(bg_lin[:, :, :3])
        fg_cam = gather.world_to_camera(fg_p.reshape(-1, 3), poses)
        bg_cam = gather.world_to_camera(bg_p4[:, :, :3].reshape(-1, 3), poses)
        inv_r = bg_p4[:, :, 3].reshape(-1, 1).unsqueeze(0).repeat(nv, 1, 1)
        bg_cam = torch.cat((bg_cam, inv_r), dim=-1)                # model.py:454-464
        fg_rgb, fg_sigma = _predict(params, fg_name, fg_cam, dir_cam, fg_world, fg_local, B, N, nv)
        bg_rgb, bg_sigma = _predict(params, bg_name, bg_cam, dir_cam, bg_world, bg_local, B, N, nv)
        wb = False if out_depth else white_bkgd
        fg_c, fg_acc, fg_w, lam, fg_depth = compositing.neo_composite(fg_rgb, fg_sigma, fg_t, d, True, far, wb)
        bg_c, bg_acc, bg_w, _, bg_depth = compositing.neo_composite(bg_rgb, bg_sigma, bg_s, d, False, None, wb)
        rgb = fg_c + lam * bg_c
        if out_depth:
            out.append((rgb, fg_c, bg_c, fg_acc, lam, fg_depth + lam.squeeze(-1) * bg_depth))
        else:
            fg_sd = 0.5 * (fg_t[..., 1:] + fg_t[..., :-1])
            fg_sd = torch.cat([fg_sd, (fg_sd[:, -1] + (fg_sd[:, -1] - fg_sd[:, -2])).unsqueeze(-1)], dim=-1)
            bg_sd = 0.5 * (bg_s[..., 1:] + bg_s[..., :-1])
            bg_sd = torch.cat([bg_sd, bg_s[..., -1].unsqueeze(-1)], dim=-1)
            out.append((rgb, fg_w, bg_w, fg_sd, bg_sd, bg_acc))
        extra.append(dict(fg_t=fg_t, bg_s=bg_s, fg_sigma=fg_sigma, bg_sigma=bg_sigma, fg_rgb=fg_rgb,
                          bg_rgb=bg_rgb, fg_w=fg_w, bg_w=bg_w, far=far))
    return (out, extra) if keep else out


_WHOLE = ("src_imgs", "src_poses", "src_focal", "src_c")


def render_chunked(params, rays, scene, chunk, **kw):
    """The caller's chunk loop (neo360/model.py:861-907): per-ray keys are
    sliced, src_* passed whole; keeps level-1 rgb ([1][0]) and depth ([1][5])."""
    B = rays["rays_o"].shape[0]
    rgb, depth = [], []
    for i in range(0, B, chunk):
        part = {k: (v if k in _WHOLE else v[i:i + chunk]) for k, v in rays.items()}
        res = render(params, part, scene, out_depth=True, **kw)
        rgb.append(res[1][0])
        depth.append(res[1][5])
    return torch.cat(rgb, 0), torch.cat(depth, 0)
