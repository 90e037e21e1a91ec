"""Oracle: Mip-NeRF 360 forward (proposal MLPs + NeRF MLP), deterministic inference path
(`randomized=False`, `is_train=False`).  Test infrastructure (see oracle/__init__.py).

Restates models/mipnerf360/helper.py and models/mipnerf360/model.py:30-365.  The one
deliberate difference in FORM: the reference obtains the contraction Jacobian with
functorch (`vmap(jacrev(_contract))`, helper.py:48-58); here it is the closed form
J = a I + (x x^T)(2/|x|^3 - 2a/|x|^2), a = (2|x|-1)/|x|^2 (identity inside the unit ball)
— the same matrix up to fp32 rounding (pinned against the reference's autograd result by
the golden fixtures).
"""
import itertools

import numpy as np
import torch
import torch.nn.functional as F

EPS = 1.1920929e-07


# ---- geodesic basis (helper.py:396-531) ---------------------------------------------------

def _sq_dist(m0, m1=None):
    m1 = m0 if m1 is None else m1
    n0, n1 = np.sum(m0 ** 2, 0), np.sum(m1 ** 2, 0)
    return np.maximum(0, n0[:, None] + n1[None, :] - 2 * m0.T @ m1)


def icosahedron_basis(subdivision=2, eps=1e-4):
    """(3,21) fp32 basis: icosahedron tesselated `subdivision` times, duplicate vertices
    merged, mirror-symmetric halves removed, columns reversed xyz->zyx (helper.py:457-531)."""
    a = (np.sqrt(5) + 1) / 2
    verts = np.array([(-1, 0, a), (1, 0, a), (-1, 0, -a), (1, 0, -a), (0, a, 1), (0, a, -1), (0, -a, 1), (0, -a, -1),
                      (a, 1, 0), (-a, 1, 0), (a, -1, 0), (-a, -1, 0)]) / np.sqrt(a + 2)
    faces = np.array([(0, 4, 1), (0, 9, 4), (9, 5, 4), (4, 5, 8), (4, 8, 1), (8, 10, 1), (8, 3, 10), (5, 3, 8), (5, 2, 3),
                      (2, 7, 3), (7, 10, 3), (7, 6, 10), (7, 11, 6), (11, 0, 6), (0, 1, 6), (6, 1, 10), (9, 0, 11),
                      (9, 11, 2), (9, 2, 5), (7, 2, 11)])
    v = subdivision
    bary = np.array([(i, j, v - (i + j)) for i in range(v + 1) for j in range(v + 1 - i)]) / v
    pts = []
    for f in faces:
        p = bary @ verts[f, :]
        pts.append(p / np.sqrt(np.sum(p ** 2, 1, keepdims=True)))
    pts = np.concatenate(pts, 0)
    first = np.array([np.min(np.argwhere(d <= eps)) for d in _sq_dist(pts.T)])
    pts = pts[np.unique(first), :]
    mirror = _sq_dist(pts.T, -pts.T) < eps
    pts = pts[np.any(np.triu(mirror), 1), :]
    return torch.from_numpy(pts[:, ::-1].copy().T).to(dtype=torch.float32)


# ---- ray casting (helper.py:278-370) ---------------------------------------------------------

def conical_frustum_gaussians(tdist, origins, directions, radii):
    """Interval [t0,t1] of a cone -> (mean (B,n,3), full covariance (B,n,3,3)).
    Follows helper.py:278-370 (ray_shape='cone', diag=False)."""
    t0, t1 = tdist[..., :-1], tdist[..., 1:]
    mu, hw = (t0 + t1) / 2, (t1 - t0) / 2
    denom = (3 * mu ** 2 + hw ** 2).clip(min=EPS)
    t_mean = mu + (2 * mu * hw ** 2) / denom
    t_var = (hw ** 2) / 3 - (4 / 15) * hw ** 4 * (12 * mu ** 2 - hw ** 2) / denom ** 2
    r_var = (mu ** 2) / 4 + (5 / 12) * hw ** 2 - (4 / 15) * (hw ** 4) / denom
    r_var = r_var * radii ** 2
    d = directions
    mean = d[..., None, :] * t_mean[..., None]
    d_mag_sq = torch.sum(d ** 2, dim=-1, keepdim=True).clip(min=1e-10)
    d_outer = d[..., :, None] * d[..., None, :]
    null_outer = torch.eye(3) - d[..., :, None] * (d / d_mag_sq)[..., None, :]
    cov = t_var[..., None, None] * d_outer[..., None, :, :] + r_var[..., None, None] * null_outer[..., None, :, :]
    return mean + origins[..., None, :], cov


# ---- contraction + integrated positional encoding (helper.py:33-88) -----------------------------

def contract(mean, cov):
    """z = x inside the unit ball, (2|x|-1)/|x|^2 x outside; cov -> J cov J^T (helper.py:33-66)."""
    m = torch.sum(mean ** 2, dim=-1, keepdim=True).clip(min=1e-32)
    a = (2 * torch.sqrt(m) - 1) / m
    inside = m <= 1
    z = torch.where(inside, mean, a * mean)
    outer = mean[..., :, None] * mean[..., None, :]
    coef = 2 / (m * torch.sqrt(m)) - 2 * a / m
    J = a[..., None] * torch.eye(3) + coef[..., None] * outer
    J = torch.where(inside[..., None], torch.eye(3).expand_as(J), J)
    return z, J @ cov @ J.transpose(-1, -2)


def lift_and_diagonalize(mean, cov, basis):
    """helper.py:70-73."""
    return mean @ basis, torch.sum(basis[None, None, ...] * (cov @ basis), dim=-2)


def integrated_pos_enc(mean, var, min_deg, max_deg):
    """exp(-var 4^k / 2) * sin(mean 2^k [+ pi/2]), k-major (helper.py:77-88, :102-103)."""
    scales = 2 ** torch.arange(min_deg, max_deg).type_as(mean)
    shape = list(mean.shape[:-1]) + [-1]
    sm = torch.reshape(mean[..., None, :] * scales[:, None], shape)
    sv = torch.reshape(var[..., None, :] * scales[:, None] ** 2, shape)
    return torch.exp(-0.5 * torch.cat([sv, sv], dim=-1)) * torch.sin(torch.cat([sm, sm + 0.5 * np.pi], dim=-1))


def dir_enc(x, deg=4):
    """helper.py:92-99 with append_identity=True."""
    scales = 2 ** torch.arange(0, deg).type_as(x)
    xb = torch.reshape(x[..., None, :] * scales[:, None], x.shape[:-1] + (-1,))
    return torch.cat([x, torch.sin(torch.cat([xb, xb + 0.5 * np.pi], dim=-1))], dim=-1)


# ---- MLP (model.py:30-176) ----------------------------------------------------------------------

def mlp(params, prefix, basis, means, covs, viewdirs, depth, rgb_branch, skip=4, x0=None):
    """MipNeRF360MLP.forward: density (B,n), rgb (B,n,3) (zeros when the rgb branch is disabled).
    x0 (B,n,504): evaluate the MLP on GIVEN encodings (the encodings carry no gradient; a gradient test that wants to separate the
    fp32 conditioning of sin(2^k x) from the MLP's arithmetic feeds both sides the same rows)."""
    if x0 is None:
        z, c = contract(means, covs)
        lm, lv = lift_and_diagonalize(z, c, basis)
        x0 = integrated_pos_enc(lm, lv, 0, 12)
    h = x0
    for i in range(depth):
        h = torch.relu(F.linear(h, params["%spts_linear.%d.weight" % (prefix, i)], params["%spts_linear.%d.bias" % (prefix, i)]))
        if i % skip == 0 and i > 0:
            h = torch.cat([h, x0], dim=-1)
    raw = F.linear(h, params[prefix + "density_layer.weight"], params[prefix + "density_layer.bias"])[..., 0]
    density = F.softplus(raw + (-1.0))
    if not rgb_branch:
        return density, torch.zeros_like(means)
    bott = F.linear(h, params[prefix + "bottleneck_layer.weight"], params[prefix + "bottleneck_layer.bias"])
    de = dir_enc(viewdirs)
    de = torch.broadcast_to(de[..., None, :], bott.shape[:-1] + (de.shape[-1],))
    v = torch.relu(F.linear(torch.cat([bott, de], dim=-1), params[prefix + "views_linear.0.weight"],
                            params[prefix + "views_linear.0.bias"]))
    rgb = torch.sigmoid(F.linear(v, params[prefix + "rgb_layer.weight"], params[prefix + "rgb_layer.bias"]))
    return density, rgb * (1 + 2 * 0.001) - 0.001


# ---- proposal resampling (helper.py:154-396) ------------------------------------------------------

def max_dilate_weights(t, w, dilation, domain):
    """helper.py:154-204 with renormalize=True."""
    p = w / torch.clip(t[..., 1:] - t[..., :-1], min=EPS)
    t0 = t[..., :-1] - dilation
    t1 = t[..., 1:] + dilation
    td = torch.sort(torch.cat([t, t0, t1], dim=-1), dim=-1).values
    td = torch.clip(td, domain[0], domain[1])
    mask = (t0[..., None, :] <= td[..., None]) & (t1[..., None, :] > td[..., None])
    pd = torch.where(mask, p[..., None, :], torch.zeros_like(p[..., None, :])).max(dim=-1).values[..., :-1]
    wd = pd * (td[..., 1:] - td[..., :-1])
    wd = wd / torch.clip(torch.sum(wd, dim=-1, keepdim=True), min=EPS)
    return td, wd


def _sorted_interp(x, xp, fp):
    """helper.py:219-234 (mask / max / min form)."""
    mask = x[..., None, :] >= xp[..., :, None]

    def pick(v, lo):
        if lo:
            return torch.max(torch.where(mask, v[..., None], v[..., :1, None]), dim=-2).values
        return torch.min(torch.where(~mask, v[..., None], v[..., -1:, None]), dim=-2).values

    fp0, fp1, xp0, xp1 = pick(fp, True), pick(fp, False), pick(xp, True), pick(xp, False)
    off = torch.clip(torch.nan_to_num((x - xp0) / (xp1 - xp0), 0), 0, 1)
    return fp0 + off * (fp1 - fp0)


def sample_intervals(t, w_logits, n, domain, jitter=None):
    """Interval endpoints (helper.py:237-243, :337-394): softmax -> cdf -> n centre quantiles -> midpoints, mirrored ends clipped
    to the domain.  Deterministic quantiles linspace(pad, 1-pad-eps, n); with jitter (B,1) = rand * max_jitter the randomized
    single_jitter form (:358-365): linspace(0, 1 - u_max, n) + jitter."""
    w = F.softmax(w_logits, dim=-1)
    cw = torch.cumsum(w[..., :-1], dim=-1).clip(max=1.0)
    lead = cw.shape[:-1] + (1,)
    cw0 = torch.cat([torch.zeros(lead).type_as(cw), cw, torch.ones(lead).type_as(cw)], dim=-1)
    if jitter is None:
        pad = 1 / (2 * n)
        u = torch.linspace(pad, 1 - pad - EPS, n)
        u = torch.broadcast_to(u, t.shape[:-1] + (n,)).type_as(t)
    else:
        u_max = EPS + (1 - EPS) / n
        u = (torch.linspace(0, 1 - u_max, n) + jitter).type_as(t)
    centers = _sorted_interp(u, cw0, t)
    mid = (centers[..., 1:] + centers[..., :-1]) / 2
    first = torch.clip(2 * centers[..., :1] - mid[..., :1], min=domain[0])
    last = torch.clip(2 * centers[..., -1:] - mid[..., -1:], max=domain[1])
    return torch.cat([first, mid, last], dim=-1)


def alpha_weights(density, tdist, dirs):
    """helper.py:246-275 with opaque_background=True (last interval infinitely wide)."""
    delta = (tdist[..., 1:] - tdist[..., :-1]) * torch.norm(dirs[..., None, :], dim=-1)
    dd = density * delta
    dd = torch.cat([dd[..., :-1], torch.full_like(dd[..., -1:], torch.inf)], dim=-1)
    alpha = 1 - torch.exp(-dd)
    trans = torch.exp(-torch.cat([torch.zeros_like(dd[..., :1]), torch.cumsum(dd[..., :-1], dim=-1)], dim=-1))
    return alpha * trans


# ---- the model (model.py:199-365) -------------------------------------------------------------------

def render(params, batch, train_frac, near, far, num_prop_samples=64, num_nerf_samples=32, num_levels=3,
           dilation_multiplier=0.5, dilation_bias=0.0025, anneal_slope=10, bg_rgb=1.0, basis=None, jitters=None, sdist_given=None,
           x0_given=None):
    """MipNeRF360.forward(batch, train_frac, False, False, near, far) -> (renderings, ray_history).
    jitters: per level (B,1) = rand * max_jitter, the randomized call's single jitter (helper.py:358-365).
    sdist_given: per level (B,n+1) interval endpoints to evaluate at instead of resampling (the sample positions of the
    implementation under test; they carry no gradient, model.py:308-309).  x0_given: per level (B,n,504) encodings (see mlp)."""
    basis = icosahedron_basis() if basis is None else basis
    o, d, vd, radii = batch["rays_o"], batch["rays_d"], batch["viewdirs"], batch["radii"]
    B = o.shape[0]
    s_near, s_far = 1 / near, 1 / far
    sdist = torch.cat([torch.full((B, 1), 0.0), torch.full((B, 1), 1.0)], dim=-1)
    weights = torch.ones(B, 1)
    prod = 1
    renderings, history = [], []
    for lvl in range(num_levels):
        is_prop = lvl < num_levels - 1
        n = num_prop_samples if is_prop else num_nerf_samples
        dilation = dilation_bias + dilation_multiplier * (1.0 - 0.0) / prod
        prod *= n
        if lvl > 0:
            sdist, weights = max_dilate_weights(sdist, weights, dilation, (0.0, 1.0))
            sdist, weights = sdist[..., 1:-1], weights[..., 1:-1]
        anneal = (anneal_slope * train_frac) / ((anneal_slope - 1) * train_frac + 1) if anneal_slope > 0 else 1.0
        logits = torch.where(sdist[..., 1:] > sdist[..., :-1], anneal * torch.log(weights + 0.0),
                             torch.full_like(weights, -torch.inf))
        if sdist_given is not None:
            sdist = sdist_given[lvl]
        else:
            sdist = sample_intervals(sdist, logits, n, (0.0, 1.0), None if jitters is None else jitters[lvl])
        sdist = sdist.detach()                                     # stop_level_grad (model.py:308-309)
        tdist = 1 / (sdist * s_far + (1 - sdist) * s_near)
        means, covs = conical_frustum_gaussians(tdist, o, d, radii)
        prefix = "mlps.%d." % lvl
        density, rgb = mlp(params, prefix, basis, means, covs, vd, 4 if is_prop else 8, not is_prop,
                           x0=None if x0_given is None else x0_given[lvl])
        weights = alpha_weights(density, tdist, d)
        acc = weights.sum(dim=-1)
        out = (weights[..., None] * rgb).sum(dim=-2) + torch.clip(1 - acc[..., None], min=0) * bg_rgb
        renderings.append({"rgb": out})
        history.append(dict(density=density, rgb=rgb, sdist=sdist, weights=weights))
    return renderings, history
