"""Oracle: pixel -> ray generation, ray/AABB slab test, unit-sphere exit depth.

Test infrastructure (see oracle/__init__.py).  fp32 torch for the camera math,
fp64 NumPy for the slab test (the reference runs that one as NumPy float64).
"""
import numpy as np
import torch


def pixel_directions(H, W, focal):
    """Camera-frame direction of every pixel: ((i-W/2)/f, -(j-H/2)/f, -1).

    Follows datasets/ray_utils.py:84-104.  Integer pixel coordinates (column i,
    row j), no half-pixel offset; the meshgrid is what kornia 0.6.1
    `create_meshgrid(H, W, normalized_coordinates=False)` returns (ray_utils.py:96).
    """
    col = torch.arange(W, dtype=torch.float32)[None, :].expand(H, W)
    row = torch.arange(H, dtype=torch.float32)[:, None].expand(H, W)
    return torch.stack([(col - W / 2) / focal, -(row - H / 2) / focal, -torch.ones(H, W)], dim=-1)


def camera_rays(dirs, c2w):
    """World-frame rays for one image; follows datasets/ray_utils.py:133-176
    with output_view_dirs=True, output_radii=True.

    Returns rays_o (HW,3), viewdirs (HW,3), rays_d (HW,3), radii (HW,).
    rays_d aliases viewdirs in the reference (:163-164), so both are the
    normalised direction.  radii use the un-normalised directions, row-to-row
    differences, last row repeating the previous difference (:153-160).
    """
    c2w = torch.as_tensor(c2w, dtype=torch.float32)
    H, W, _ = dirs.shape
    world = dirs @ c2w[:, :3].T
    origin = c2w[:, 3].expand(H, W, 3)
    step = torch.sqrt(((world[:-1] - world[1:]) ** 2).sum(-1))
    step = torch.cat([step, step[-2:-1]], dim=0)
    radii = (step[..., None] * 2 / torch.sqrt(torch.tensor(12, dtype=torch.int8))).reshape(-1)
    unit = world / torch.norm(world, dim=-1, keepdim=True)
    return origin.reshape(-1, 3), unit.reshape(-1, 3), unit.reshape(-1, 3).clone(), radii


def aabb_slab(bounds, orig, direc):
    """One ray vs one axis-aligned box, float64.  Returns (hit, tmin, tmax).

    Follows datasets/ray_utils.py:34-68 (== models/neo360/helper.py:291-323):
    zero direction components become 1e-14, x then y then z slabs with the
    early-outs in that order, and a final rejection when tmin<0 or tmax<0
    (origin inside / box behind).  Misses return (False, 0, 0).
    """
    d = np.array(direc, dtype=np.float64)
    d[d == 0] = 1.0e-14
    inv = 1 / d
    neg = (inv < 0).astype(np.int64)
    lo = (bounds[neg[0]][0] - orig[0]) * inv[0]
    hi = (bounds[1 - neg[0]][0] - orig[0]) * inv[0]
    for ax in (1, 2):
        a_lo = (bounds[neg[ax]][ax] - orig[ax]) * inv[ax]
        a_hi = (bounds[1 - neg[ax]][ax] - orig[ax]) * inv[ax]
        if lo > a_hi or a_lo > hi:
            return False, 0.0, 0.0
        if a_lo > lo:
            lo = a_lo
        if a_hi < hi:
            hi = a_hi
    if lo < 0 or hi < 0:
        return False, 0.0, 0.0
    return True, lo, hi


def aabb_slab_batch(bounds, rays_o, rays_d):
    """Loop of `aabb_slab` over rays; follows datasets/ray_utils.py:17-31.
    Returns float64 arrays (hit as 0.0/1.0, tmin, tmax), like the reference."""
    bounds = np.asarray(bounds, dtype=np.float64)
    rays_o = np.asarray(rays_o, dtype=np.float64)
    rays_d = np.asarray(rays_d, dtype=np.float64)
    n = rays_o.shape[0]
    hit = np.empty(n)
    tmin = np.empty(n)
    tmax = np.empty(n)
    for k in range(n):
        hit[k], tmin[k], tmax[k] = aabb_slab(bounds, rays_o[k], rays_d[k])
    return hit, tmin, tmax


def rays_to_box_frame(rays_o, rays_d, box_from_world):
    """o' = R o + t, d' = R d with [R|t] = box_from_world (4x4, float64).
    Follows models/neo360/helper.py:325-331."""
    T = np.asarray(box_from_world, dtype=np.float64)
    o = (T[:3, :3] @ np.asarray(rays_o, dtype=np.float64).T).T + T[:3, 3]
    d = (T[:3, :3] @ np.asarray(rays_d, dtype=np.float64).T).T
    return o, d


def rays_in_box(rays_o, rays_d, bounds, box_from_world):
    """mask (bool), near (R,1), far (R,1) as float32 torch tensors.
    Follows models/neo360/helper.py:333-346."""
    o, d = rays_to_box_frame(rays_o, rays_d, box_from_world)
    hit, tmin, tmax = aabb_slab_batch(bounds, o, d)
    return torch.Tensor(hit).bool(), torch.Tensor(tmin[:, None]), torch.Tensor(tmax[:, None])


def merge_boxes(nears, fars):
    """Multi-box merge of per-box (near, far) with 0 meaning "no hit".
    Follows models/neo360/helper.py:359-373: running element-wise min that
    treats 0 as absent; mask = both non-zero."""
    all_near = torch.zeros_like(nears[0])
    all_far = torch.zeros_like(fars[0])
    for near, far in zip(nears, fars):
        all_near = torch.where((all_near == 0) | (near == 0), torch.maximum(near, all_near), torch.minimum(near, all_near))
        all_far = torch.where((all_far == 0) | (far == 0), torch.maximum(far, all_far), torch.minimum(far, all_far))
    return all_near, all_far, (all_near != 0) & (all_far != 0)


def sample_rays_in_bbox(RTs, rays_o, view_dirs):
    """models/neo360/helper.py:348-373: per object, box frame = inverse([R|T]) (get_object_rays_in_bbox :348-357),
    rays_in_box, running merge.  rays_o / view_dirs: NumPy (R,3).  Returns all_near, all_far (R,1) float32,
    bbox_mask (R,1) bool and the list of per-box hit masks."""
    nears, fars, hits = [], [], []
    for rot, tran, sca in zip(RTs["R"], RTs["T"], RTs["s"]):
        box = np.eye(4)
        box[:3, :3] = np.reshape(np.array(rot), (3, 3))
        box[:3, -1] = np.array(tran)
        hit, near, far = rays_in_box(rays_o, view_dirs, np.array(sca), np.linalg.inv(box))
        hits.append(hit)
        nears.append(near)
        fars.append(far)
    all_near, all_far, mask = merge_boxes(nears, fars)
    return all_near, all_far, mask, hits


def sphere_exit_depth(rays_o, rays_d):
    """Depth at which each ray leaves the unit sphere, (B,1).

    Follows models/neo360/helper.py:253-273: d1 = -(d.o)/(d.d) (closest
    approach), d2 = sqrt(1-|o+d1 d|^2)/|d|; asserts every ray meets the
    sphere (:271) — that assertion is the path's second "hit mask".
    Also returns the boolean per-ray check so callers can compare masks.
    """
    d1 = -(rays_d * rays_o).sum(-1, keepdim=True) / (rays_d ** 2).sum(-1, keepdim=True)
    closest = rays_o + d1 * rays_d
    inv_len = 1.0 / torch.norm(rays_d, dim=-1, keepdim=True)
    margin = 1.0 - (closest * closest).sum(-1, keepdim=True)
    ok = margin >= 0
    assert bool(torch.all(ok)), "1.0 - p_norm_sq should be greater than 0"
    return d1 + torch.sqrt(margin) * inv_len, ok
