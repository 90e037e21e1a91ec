"""Stage-level operators over the C ABI — the GPU counterparts of the reference's
helper functions, used by the modules in `models.py` and by the stage-isolated
parity tests.  Inputs and outputs are PyTorch-ROCm tensors."""
import ctypes

import torch

from . import _lib
from .context import f32, f64, get_context, ptr


def _ctx(t, ctx):
    return ctx if ctx is not None else get_context(t.device)


def get_ray_directions_and_rays(H, W, focal, c2w, ctx=None, device="cuda", ray_range=None, out=None):
    """datasets/ray_utils.py:84-104 + :133-176 in one kernel.
    c2w: (3,4) or (4,4) array-like (host).  Returns rays_o, viewdirs, rays_d (n,3), radii (n,) for the whole frame
    (n = H W) or for rays [lo, hi) of it when ray_range = (lo, hi) (a rank's shard).  `out`: tensors to write into
    (e.g. the previous frame's, so a render loop allocates nothing)."""
    ctx = ctx if ctx is not None else get_context(device)
    dev = ctx.device
    pose = torch.as_tensor(c2w, dtype=torch.float32, device="cpu")[:3, :4].contiguous()
    host = (ctypes.c_float * 12)(*pose.reshape(-1).tolist())
    lo, hi = ray_range if ray_range is not None else (0, H * W)
    n = hi - lo
    if out is not None and out[0].shape[0] == n:
        rays_o, viewdirs, rays_d, radii = out
        _lib.note_external_write()      # same tensor objects, same version counters, new contents
    else:
        rays_o = torch.empty(n, 3, device=dev)
        viewdirs = torch.empty(n, 3, device=dev)
        rays_d = torch.empty(n, 3, device=dev)
        radii = torch.empty(n, device=dev)
    _lib.check(ctx.lib.neo_raygen_range(ctx.handle, H, W, float(focal), host, int(lo), int(n), ptr(rays_o), ptr(viewdirs),
                                        ptr(rays_d), ptr(radii), ctx.stream()))
    return rays_o, viewdirs, rays_d, radii


def bbox_intersection_batch(bounds, rays_o, rays_d, ctx=None):
    """datasets/ray_utils.py:17-31 (float64).  bounds (2,3) host array-like; rays in the
    box frame, (R,3) float64 device tensors.  Returns hit (uint8), tmin, tmax (float64)."""
    rays_o, rays_d = f64(rays_o, "rays_o"), f64(rays_d, "rays_d")
    ctx = _ctx(rays_o, ctx)
    R = rays_o.shape[0]
    b = torch.as_tensor(bounds, dtype=torch.float64, device="cpu").reshape(6)
    host = (ctypes.c_double * 6)(*b.tolist())
    hit = torch.empty(R, dtype=torch.uint8, device=rays_o.device)
    tmin = torch.empty(R, dtype=torch.float64, device=rays_o.device)
    tmax = torch.empty(R, dtype=torch.float64, device=rays_o.device)
    _lib.check(ctx.lib.neo_aabb_intersect(ctx.handle, host, ptr(rays_o), ptr(rays_d), R, ptr(hit), ptr(tmin),
                                          ptr(tmax), ctx.stream()))
    return hit, tmin, tmax


def sample_rays_in_bbox(RTs, rays_o, view_dirs, ctx=None, return_per_box=False):
    """models/neo360/helper.py:359-373 (with get_object_rays_in_bbox :348-357 and get_rays_in_bbox :333-346 inside):
    RTs = dict(R=[3x3 or 9], T=[3], s=[(2,3) bounds]) per object; rays are moved into every box frame in float64,
    slab-tested, and merged with 0 as the "no hit" sentinel.  rays_o / view_dirs: (R,3) device tensors (any float
    dtype; the reference's NumPy arrays are promoted to float64 the same way).  Returns all_near (R,1), all_far (R,1)
    float32 and bbox_mask (R,1) bool, as the reference; with return_per_box also the per-box hit masks (n,R)."""
    import numpy as np
    rays_o, view_dirs = f64(rays_o, "rays_o"), f64(view_dirs, "view_dirs")
    ctx = _ctx(rays_o, ctx)
    R = rays_o.shape[0]
    mats, bounds = [], []
    for rot, tran, sca in zip(RTs["R"], RTs["T"], RTs["s"]):
        box = np.eye(4)                                         # helper.py:352-356, verbatim order of operations
        box[:3, :3] = np.reshape(np.array(rot), (3, 3))
        box[:3, -1] = np.array(tran)
        mats.append(np.linalg.inv(box))
        bounds.append(np.asarray(sca, dtype=np.float64).reshape(6))
    n = len(mats)
    dev = rays_o.device
    if n == 0:
        # a scene without objects: the reference's loops never run and its zero-initialised outputs come back
        # (helper.py:359-373): near = far = 0, mask all False
        zero = torch.zeros(R, 1, device=dev)
        mask = torch.zeros(R, 1, dtype=torch.bool, device=dev)
        if return_per_box:
            return zero, zero.clone(), mask, torch.empty(0, R, dtype=torch.uint8, device=dev)
        return zero, zero.clone(), mask
    hm = (ctypes.c_double * (16 * n))(*np.stack(mats).astype(np.float64).reshape(-1).tolist())
    hb = (ctypes.c_double * (6 * n))(*np.stack(bounds).reshape(-1).tolist())
    near = torch.empty(R, 1, device=dev)
    far = torch.empty(R, 1, device=dev)
    mask = torch.empty(R, 1, dtype=torch.uint8, device=dev)
    per_box = torch.empty(n, R, dtype=torch.uint8, device=dev) if return_per_box else None
    _lib.check(ctx.lib.neo_aabb_multi(ctx.handle, n, hm, hb, ptr(rays_o), ptr(view_dirs), R, ptr(per_box), ptr(near),
                                      ptr(far), ptr(mask), ctx.stream()))
    if return_per_box:
        return near, far, mask.bool(), per_box
    return near, far, mask.bool()


def intersect_sphere(rays_o, rays_d, ctx=None, check=True):
    """models/neo360/helper.py:253-273.  Returns far (R,1) and the per-ray hit mask (uint8).
    Raises AssertionError like the reference when a ray misses the unit sphere."""
    rays_o, rays_d = f32(rays_o, "rays_o"), f32(rays_d, "rays_d")
    ctx = _ctx(rays_o, ctx)
    R = rays_o.shape[0]
    far = torch.empty(R, 1, device=rays_o.device)
    ok = torch.empty(R, dtype=torch.uint8, device=rays_o.device)
    _lib.check(ctx.lib.neo_intersect_sphere(ctx.handle, ptr(rays_o), ptr(rays_d), R, ptr(far), ptr(ok), ctx.stream()))
    if check and (ctx.poll_flags() & 1):
        raise AssertionError("1.0 - p_norm_sq should be greater than 0")
    return far, ok


def pos_enc(x, min_deg, max_deg, ctx=None):
    """neo360/helper.py:121-125 == vanilla_nerf/helper.py:445-449."""
    x = f32(x, "x")
    ctx = _ctx(x, ctx)
    C = x.shape[-1]
    n = x.numel() // C
    out = torch.empty(*x.shape[:-1], C * (2 * (max_deg - min_deg) + 1), device=x.device)
    _lib.check(ctx.lib.neo_pos_enc(ctx.handle, ptr(x), n, C, min_deg, max_deg, ptr(out), ctx.stream()))
    return out


def resample(t_prev, weights, n_new, descending=False, ctx=None):
    """sorted_piecewise_constant_pdf + the sort of sample_pdf, with the callers'
    slicing (bins = midpoints of t_prev, pdf weights = weights[:,1:-1])."""
    t_prev, weights = f32(t_prev, "t_prev"), f32(weights, "weights")
    ctx = _ctx(t_prev, ctx)
    R, n_prev = t_prev.shape
    out = torch.empty(R, n_prev + n_new, device=t_prev.device)
    _lib.check(ctx.lib.neo_resample(ctx.handle, ptr(t_prev), ptr(weights), R, n_prev, n_new, int(descending),
                                    ptr(out), ctx.stream()))
    return out


def composite(mode, rgbsigma, t, rays_d=None, t_far=None, white_bkgd=False, ctx=None):
    """volumetric_rendering; mode 0 vanilla, 1 NeO-360 inside, 2 NeO-360 outside.
    Returns dict(rgb, acc, depth, weights, bg_lambda)."""
    rgbsigma, t = f32(rgbsigma, "rgbsigma"), f32(t, "t")
    ctx = _ctx(t, ctx)
    R, N = t.shape
    dev = t.device
    rays_d = f32(rays_d, "rays_d") if rays_d is not None else None
    t_far = f32(t_far, "t_far") if t_far is not None else None
    rgb = torch.empty(R, 3, device=dev)
    acc = torch.empty(R, device=dev)
    depth = torch.empty(R, device=dev)
    w = torch.empty(R, N, device=dev)
    lam = torch.empty(R, 1, device=dev) if mode == 1 else None
    _lib.check(ctx.lib.neo_composite(ctx.handle, mode, ptr(rgbsigma), ptr(t), ptr(rays_d), ptr(t_far), R, N,
                                     int(bool(white_bkgd)), ptr(rgb), ptr(acc), ptr(depth), ptr(w), ptr(lam),
                                     ctx.stream()))
    return dict(rgb=rgb, acc=acc, depth=depth, weights=w, bg_lambda=lam)


def mip_resample(s_prev, w_prev, n, near, far, dilate=False, dilation=0.0, anneal=1.0, ctx=None):
    """One Mip-NeRF 360 proposal-resampling step (mipnerf360/model.py:258-310).  Returns sdist, tdist (R,n+1)."""
    s_prev, w_prev = f32(s_prev, "s_prev"), f32(w_prev, "w_prev")
    ctx = _ctx(s_prev, ctx)
    R, n_prev = w_prev.shape
    sdist = torch.empty(R, n + 1, device=s_prev.device)
    tdist = torch.empty(R, n + 1, device=s_prev.device)
    _lib.check(ctx.lib.neo_mip_resample(ctx.handle, ptr(s_prev), ptr(w_prev), R, n_prev, int(bool(dilate)),
                                        float(dilation), float(anneal), n, float(near), float(far), ptr(sdist),
                                        ptr(tdist), ctx.stream()))
    return sdist, tdist


def mip_composite(rgbdens, tdist, rays_d, bg=1.0, ctx=None):
    """compute_alpha_weights(opaque_background=True) + volumetric_rendering (mipnerf360/helper.py:246-274)."""
    rgbdens, tdist, rays_d = f32(rgbdens), f32(tdist), f32(rays_d)
    ctx = _ctx(tdist, ctx)
    R, n1 = tdist.shape
    w = torch.empty(R, n1 - 1, device=tdist.device)
    rgb = torch.empty(R, 3, device=tdist.device)
    _lib.check(ctx.lib.neo_mip_composite(ctx.handle, ptr(rgbdens), ptr(tdist), ptr(rays_d), R, n1 - 1, float(bg), ptr(w),
                                         ptr(rgb), ctx.stream()))
    return w, rgb
