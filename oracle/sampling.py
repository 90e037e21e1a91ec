"""Oracle: along-ray sample placement (deterministic, `randomized=False` path).

Test infrastructure (see oracle/__init__.py).
"""
import torch


def points_on_rays(t, origins, directions):
    """o + t*d, separate multiply and add.  vanilla_nerf/helper.py:20-21,
    neo360/helper.py:20-21."""
    return origins[..., None, :] + t[..., None] * directions[..., None, :]


def unit_edges(n):
    """The n+1 edges linspace(0,1,n+1) both samplers start from
    (vanilla_nerf/helper.py:425, neo360/helper.py:36)."""
    return torch.linspace(0.0, 1.0, n + 1)


# ----------------------------------------------------------------------------
# vanilla NeRF
# ----------------------------------------------------------------------------

def vanilla_level0(rays_o, dirs, n, near, far):
    """t (B,n+1) and points (B,n+1,3) for the coarse level, scalar near/far.
    Follows vanilla_nerf/helper.py:415-442 (lindisp=False, randomized=False)."""
    s = unit_edges(n)
    t = near * (1.0 - s) + far * s
    t = torch.broadcast_to(t, (rays_o.shape[0], n + 1))
    return t, points_on_rays(t, rays_o, dirs)


# ----------------------------------------------------------------------------
# NeO-360 (NeRF++ inside / outside the unit sphere)
# ----------------------------------------------------------------------------

def neo_fg_level0(rays_o, rays_d, n, near, far):
    """Inside-sphere samples: t = near(1-s)+far s with per-ray near/far (B,1).
    Follows neo360/helper.py:24-57 (in_sphere=True)."""
    s = unit_edges(n)
    t = near * (1.0 - s) + far * s
    return t, points_on_rays(t, rays_o, rays_d)


def neo_bg_level0(rays_o, rays_d, n, far, far_uncontracted=3.0):
    """Outside-sphere samples.  Follows neo360/helper.py:24-75 (in_sphere=False):
    s ascending 0..1 -> linear depth far(1-s)+far_unc*s; both flipped so the
    inverse radius s runs 1 -> 0; returns (s_desc (B,n+1), pts4 (B,n+1,4),
    pts_linear (B,n+1,3))."""
    s = torch.broadcast_to(unit_edges(n), (rays_o.shape[0], n + 1))
    t_lin = far * (1.0 - s) + far_uncontracted * s
    s_desc = torch.flip(s, dims=[-1])
    t_lin = torch.flip(t_lin, dims=[-1])
    return s_desc, inverted_sphere_points(rays_o, rays_d, s_desc), points_on_rays(t_lin, rays_o, rays_d)


def inverted_sphere_points(rays_o, rays_d, inv_r):
    """NeRF++ inverted-sphere parameterisation, (B,N) -> (B,N,4) = (x',y',z',1/r).

    Follows neo360/helper.py:401-451: the ray's unit-sphere exit point is rotated
    about axis (o x p_sphere) by asin(|p_mid|) - asin(|p_mid|*inv_r) (Rodrigues),
    re-normalised (+1e-10), and 1/r appended.  Asserts the ray meets the sphere (:426).
    """
    shape = list(inv_r.shape) + [3]
    o = rays_o[..., None, :].expand(shape)
    d = rays_d[..., None, :].expand(shape)
    d1 = -(d * o).sum(-1, keepdim=True) / (d ** 2).sum(-1, keepdim=True)
    p_mid = o + d1 * d
    r_mid = torch.norm(p_mid, dim=-1, keepdim=True)
    inv_len = 1.0 / torch.norm(d, dim=-1, keepdim=True)
    assert bool(torch.all(1.0 - r_mid * r_mid >= 0)), "1.0 - p_mid_norm * p_mid_norm should be greater than 0"
    d2 = torch.sqrt(1.0 - r_mid * r_mid) * inv_len
    p_sph = o + (d1 + d2) * d
    axis = torch.cross(o, p_sph, dim=-1)
    axis = axis / torch.norm(axis, dim=-1, keepdim=True)
    ang = torch.asin(r_mid) - torch.asin(r_mid * inv_r[..., None])
    turned = (
        p_sph * torch.cos(ang)
        + torch.cross(axis, p_sph, dim=-1) * torch.sin(ang)
        + axis * (axis * p_sph).sum(-1, keepdim=True) * (1.0 - torch.cos(ang))
    )
    turned = turned / (torch.norm(turned, dim=-1, keepdim=True) + 1e-10)
    return torch.cat((turned, inv_r.unsqueeze(-1)), dim=-1)


# ----------------------------------------------------------------------------
# hierarchical (inverse-CDF) resampling — shared by vanilla and NeO-360
# ----------------------------------------------------------------------------

def piecewise_constant_samples(bins, weights, n_new, float_min_eps=2 ** -32, u=None):
    """n_new deterministic inverse-CDF samples per ray.

    Follows neo360/helper.py:174-215 == vanilla_nerf/helper.py:567-607
    (randomized=False): weights padded so their sum >= 1e-5; pdf; cdf =
    [0, min(1, cumsum(pdf[:-1])), 1]; u = linspace(0, 1-2^-32, n_new); the
    bracketing bin/cdf pair is found with the reference's mask / max / min
    formulation — NOT a searchsorted — so it is valid for the descending bins
    of the background branch too; t = clip(nan->0((u-cdf0)/(cdf1-cdf0)),0,1).
    """
    total = weights.sum(dim=-1, keepdim=True)
    pad = torch.fmax(torch.zeros_like(total), 1e-5 - total)
    weights = weights + pad / weights.shape[-1]
    total = total + pad
    pdf = weights / total
    inner = torch.fmin(torch.ones_like(pdf[..., :-1]), torch.cumsum(pdf[..., :-1], dim=-1))
    lead = list(inner.shape[:-1]) + [1]
    cdf = torch.cat([torch.zeros(lead), inner, torch.ones(lead)], dim=-1)
    if u is None:       # randomized=False; randomized=True passes its uniform draws (helper.py:195-196)
        u = torch.linspace(0.0, 1.0 - float_min_eps, n_new)
        u = torch.broadcast_to(u, list(cdf.shape[:-1]) + [n_new])
    ge = u[..., None, :] >= cdf[..., :, None]

    def below(x):  # largest x_j among those with u >= cdf_j (others read x_0)
        return (ge * x[..., None] + ~ge * x[..., :1, None]).max(dim=-2)[0]

    def above(x):  # smallest x_j among those with u < cdf_j (others read x_last)
        return (~ge * x[..., None] + ge * x[..., -1:, None]).min(dim=-2)[0]

    b0, b1, c0, c1 = below(bins), above(bins), below(cdf), above(cdf)
    frac = torch.clip(torch.nan_to_num((u - c0) / (c1 - c0), 0), 0, 1)
    return b0 + frac * (b1 - b0)


def merge_sorted(t_prev, t_new):
    """sort(cat(previous t, new samples)) ascending
    (vanilla_nerf/helper.py:614, neo360/helper.py:227)."""
    return torch.sort(torch.cat([t_prev, t_new], dim=-1), dim=-1).values


def vanilla_level1(bins, weights, rays_o, dirs, t_prev, n_new):
    """vanilla_nerf/helper.py:610-616."""
    t = merge_sorted(t_prev, piecewise_constant_samples(bins, weights, n_new))
    return t, points_on_rays(t, rays_o, dirs)


def neo_fg_level1(bins, weights, rays_o, rays_d, t_prev, n_new):
    """neo360/helper.py:218-231 (in_sphere=True)."""
    t = merge_sorted(t_prev, piecewise_constant_samples(bins, weights, n_new).detach())      # helper.py:224: .detach()
    return t, points_on_rays(t, rays_o, rays_d)


def neo_bg_level1(bins, weights, rays_o, rays_d, s_prev, n_new, far, far_uncontracted=3.0):
    """neo360/helper.py:218-249 (in_sphere=False): merged set sorted ascending,
    linear depth from the ascending set, then both flipped to descending."""
    s_asc = merge_sorted(s_prev, piecewise_constant_samples(bins, weights, n_new).detach())  # helper.py:224: .detach()
    t_lin = far * (1.0 - s_asc) + far_uncontracted * s_asc
    s_desc = torch.flip(s_asc, dims=[-1])
    t_lin = torch.flip(t_lin, dims=[-1])
    return s_desc, inverted_sphere_points(rays_o, rays_d, s_desc), points_on_rays(t_lin, rays_o, rays_d)
