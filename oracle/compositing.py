"""Oracle: along-ray alpha compositing.  Test infrastructure (oracle/__init__.py)."""
import torch

_EPS = 1e-10


def vanilla_composite(rgb, sigma, t, dirs, white_bkgd=False):
    """rgb (B,N,3), sigma (B,N,1), t (B,N), dirs (B,3) -> rgb, acc, weights, depth.

    Follows vanilla_nerf/helper.py:521-559: delta_i = t_{i+1}-t_i, last = 1e10,
    scaled by |dirs|; alpha = 1-exp(-sigma delta); w_i = alpha_i *
    prod_{j<i}(1-alpha_j+1e-10); depth = sum w t, then nan_to_num(+inf) and a
    clamp to the call's own [min,max] (:546-547).
    """
    delta = torch.cat([t[..., 1:] - t[..., :-1], torch.ones(t[..., :1].shape) * 1e10], dim=-1)
    delta = delta * torch.norm(dirs[..., None, :], dim=-1)
    alpha = 1.0 - torch.exp(-sigma[..., 0] * delta)
    trans = torch.cat([torch.ones_like(alpha[..., :1]), torch.cumprod(1.0 - alpha[..., :-1] + _EPS, dim=-1)], dim=-1)
    w = alpha * trans
    out = (w[..., None] * rgb).sum(dim=-2)
    depth = (w * t).sum(dim=-1)
    depth = torch.nan_to_num(depth, float("inf"))
    depth = torch.clamp(depth, torch.min(depth), torch.max(depth))
    acc = w.sum(dim=-1)
    if white_bkgd:
        out = out + (1.0 - acc[..., None])
    return out, acc, w, depth


def neo_composite(rgb, sigma, t, dirs, in_sphere, t_far=None, white_bkgd=False):
    """NeRF++ two-region compositing -> rgb, acc, weights, bg_lambda, depth.

    Follows neo360/helper.py:128-171.  Inside the sphere: delta = forward
    differences, last = t_far - t_last, scaled by |dirs|.  Outside: t holds the
    DEscending inverse radius, delta_i = t_i - t_{i+1}, last = 1e10, no |dirs|.
    T = cumprod(1-alpha+1e-10) INclusive; bg_lambda = T_last (inside only);
    w_i = alpha_i * T_{i-1} (T_{-1}=1); depth = sum w t.
    """
    if in_sphere:
        delta = torch.cat([t[..., 1:] - t[..., :-1], t_far - t[..., -1:]], dim=-1)
        delta = delta * torch.norm(dirs[..., None, :], dim=-1)
    else:
        delta = torch.cat([t[..., :-1] - t[..., 1:], torch.full_like(t[..., :1], 1e10)], dim=-1)
    alpha = 1.0 - torch.exp(-sigma[..., 0] * delta)
    T = torch.cumprod(1.0 - alpha + _EPS, dim=-1)
    lam = T[..., -1:] if in_sphere else None
    w = alpha * torch.cat([torch.ones_like(T[..., -1:]), T[..., :-1]], dim=-1)
    acc = w.sum(dim=-1)
    out = (w[..., None] * rgb).sum(dim=-2)
    if white_bkgd:
        out = out + (1.0 - acc[..., None])
    return out, acc, w, lam, (w * t).sum(dim=-1)
