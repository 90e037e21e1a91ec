"""Oracle: training-side pieces of the NeO-360 path.  Test infrastructure (oracle/__init__.py).

  * stratified level-0 sampling and randomized resampling, restated from neo360/helper.py:24-75, :174-249 with the
    uniform draws as explicit inputs (the reference calls torch.rand; tests/test_oracle_vs_reference.py feeds it the
    same numbers by patching torch.rand);
  * the counter-based generator those draws come from on the device: Philox4x32-10 (Salmon, Moraes, Dror, Shaw:
    "Parallel random numbers: as easy as 1, 2, 3", SC'11), restated in NumPy integer arithmetic;
  * eff_distloss of torch_efficient_distloss (requirements.txt:29, version unpinned by the reference; package absent
    here): the published algorithm (Sun et al., "Improved Direct Voxel Grid Optimization", 2022, eq. 5-7): O(N) prefix-sum
    evaluation of  interval/3 sum_i w_i^2 + sum_{i,j} w_i w_j |m_i - m_j|  for sorted m.  Call site neo360/model.py:1246-1260.
Gradients are checked against torch.autograd of these restatements (compositing: oracle.compositing; lookups: oracle.gather).
"""
import numpy as np
import torch

from . import sampling

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox_uniform(seed, stream_id, rows, cols):
    """(rows, cols) float32 uniforms in [0,1): word 0 of Philox4x32-10 with key = (seed lo, seed hi) and counter
    (row, col, stream_id, 0), top 24 bits scaled by 2^-24 (torch.rand's fp32 convention)."""
    r, c = np.meshgrid(np.arange(rows, dtype=np.uint32), np.arange(cols, dtype=np.uint32), indexing="ij")
    x0, x1 = r.copy(), c.copy()
    x2 = np.full_like(x0, np.uint32(stream_id))
    x3 = np.zeros_like(x0)
    k0, k1 = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _M0 * x0.astype(np.uint64)
            p1 = _M1 * x2.astype(np.uint64)
            y0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ x1 ^ k0
            y1 = (p1 & _MASK).astype(np.uint32)
            y2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ x3 ^ k1
            y3 = (p0 & _MASK).astype(np.uint32)
            x0, x1, x2, x3 = y0, y1, y2, y3
            k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return torch.from_numpy(((x0 >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)))


def _stratify(t_vals, u):
    """neo360/helper.py:44-51: one draw per sample between the midpoints to its neighbours."""
    mids = 0.5 * (t_vals[..., 1:] + t_vals[..., :-1])
    upper = torch.cat([mids, t_vals[..., -1:]], -1)
    lower = torch.cat([t_vals[..., :1], mids], -1)
    return lower + (upper - lower) * u


def neo_level0_randomized(far, n, u_fg, u_bg, near=1e-4):
    """Both regions' level-0 sample rows with randomized=True (helper.py:24-75): fg_t ascending (R,n+1), bg_s = the
    jittered inverse radii flipped to descending.  far (R,1); u_* (R,n+1)."""
    e = sampling.unit_edges(n)
    nr = torch.full_like(far, near)
    fg = _stratify(nr * (1.0 - e) + far * e, u_fg)
    bg = _stratify(torch.broadcast_to(e, (far.shape[0], n + 1)), u_bg)
    return fg, torch.flip(bg, dims=[-1])


def resample_randomized(t_prev, weights, u, descending=False):
    """sample_pdf with randomized=True (helper.py:218-231): bins = midpoints of t_prev, pdf weights = weights[:,1:-1]
    (the callers' slicing, model.py:308-318), draws u (R, n_new); merged and sorted (descending for the bg branch)."""
    mids = 0.5 * (t_prev[..., 1:] + t_prev[..., :-1])
    new = sampling.piecewise_constant_samples(mids, weights[..., 1:-1], u.shape[-1], u=u).detach()      # helper.py:224
    merged = torch.sort(torch.cat([t_prev, new], dim=-1), dim=-1).values
    return torch.flip(merged, dims=[-1]) if descending else merged


def eff_distloss(w, m, interval):
    """torch_efficient_distloss.eff_distloss forward (differentiable torch ops): mean over rays of
    interval/3 sum w^2 + 2 sum_{i>=1} (w_i m_i W_{i-1} - w_i WM_{i-1}), W / WM inclusive prefix sums of w / w m."""
    loss_uni = (1.0 / 3.0) * (interval * w.pow(2)).sum(dim=-1).mean()
    wm = w * m
    w_cumsum, wm_cumsum = w.cumsum(dim=-1), wm.cumsum(dim=-1)
    loss_bi = 2.0 * (wm[..., 1:] * w_cumsum[..., :-1] - w[..., 1:] * wm_cumsum[..., :-1]).sum(dim=-1).mean()
    return loss_bi + loss_uni


def distloss_bruteforce(w, m, interval):
    """The O(N^2) definition the prefix-sum form evaluates (m ascending): checks the restatement itself."""
    pair = (w[..., :, None] * w[..., None, :] * (m[..., :, None] - m[..., None, :]).abs()).sum(dim=(-1, -2))
    return (pair + (1.0 / 3.0) * interval * w.pow(2).sum(dim=-1)).mean()
