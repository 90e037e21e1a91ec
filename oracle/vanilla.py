"""Oracle: vanilla NeRF coarse+fine render.  Test infrastructure (oracle/__init__.py)."""
import torch

from . import compositing, encoding, mlp, sampling


def render(params, rays, near, far, n_coarse=64, n_fine=128, white_bkgd=False, keep=False, samples=None, sigma_noise=None):
    """[(rgb (B,3), acc (B,), depth (B,))] x 2 — the return value of
    NeRF.forward (vanilla_nerf/model.py:154-216) for randomized=False.

    Points are cast along `viewdirs` (:161,:177); compositing scales by
    |rays_d| (:207-212).  `keep=True` appends per-level intermediates
    (t, sigma, rgb, weights) for stage-level parity tests.
    samples = (t0 (B,n_coarse+1), t1 (B,n_coarse+1+n_fine)): evaluate at GIVEN sample positions instead of the
    deterministic ones (the randomized=True branches :163-166 / helper.py:431-436, :589-590 draw them; positions carry no
    gradient, helper.py:612).  sigma_noise = (u0, u1) x noise_std already applied: added to the raw density (:194-195).
    """
    o, vd, rd = rays["rays_o"], rays["viewdirs"], rays["rays_d"]
    dir_enc = encoding.pos_enc(vd, 0, 4)
    out, extra = [], []
    t = w = None
    for level, prefix in enumerate(("coarse_mlp.", "fine_mlp.")):
        if samples is not None:
            t = samples[level]
            pts = sampling.points_on_rays(t, o, vd)
        elif level == 0:
            t, pts = sampling.vanilla_level0(o, vd, n_coarse, near, far)
        else:
            mids = 0.5 * (t[..., 1:] + t[..., :-1])
            t, pts = sampling.vanilla_level1(mids, w[..., 1:-1], o, vd, t, n_fine)
        raw_rgb, raw_sigma = mlp.vanilla_mlp(params, prefix, encoding.pos_enc(pts, 0, 10), dir_enc)
        if sigma_noise is not None:
            raw_sigma = raw_sigma + sigma_noise[level].reshape(raw_sigma.shape)
        rgb = mlp.colour_activation(raw_rgb)
        sigma = mlp.density_activation(raw_sigma)
        comp, acc, w, depth = compositing.vanilla_composite(rgb, sigma, t, rd, white_bkgd)
        out.append((comp, acc, depth))
        extra.append(dict(t=t, sigma=sigma, rgb=rgb, weights=w))
    return (out, extra) if keep else out


def render_chunked(params, rays, near, far, chunk, **kw):
    """The caller's chunk loop (vanilla_nerf/model.py:336-363): slices every
    per-ray key, keeps level-1 rgb and depth, concatenates."""
    B = rays["rays_o"].shape[0]
    rgb, depth = [], []
    for i in range(0, B, chunk):
        part = {k: v[i:i + chunk] for k, v in rays.items()}
        res = render(params, part, near, far, **kw)
        rgb.append(res[1][0])
        depth.append(res[1][2])
    return torch.cat(rgb, 0), torch.cat(depth, 0)
