"""Multi-GPU: rays of one frame are independent, so a frame is sharded across
ranks by contiguous ranges of WHOLE reference chunks (NeO-360 results depend on
chunk membership) and reassembled with ONE collective: an RCCL
all_gather_into_tensor of a packed (rays, C) fp32 tile per rank (~0.8 MB at 8
GPUs: latency-bound over xGMI, so a single direct all-gather, not a ring of
small messages).  This replaces, for the single-frame case, the reference's only
collective call site, LitModel.alter_gather_cat (models/interface.py:30-50).
"""
import torch


def shard_counts(n_rays, world, unit=1024):
    """Rays per rank: whole `unit`-sized chunks dealt as evenly as possible
    (300 chunks over 8 ranks -> 38,38,38,38,37,37,37,37); the last chunk may be short."""
    n_units = (n_rays + unit - 1) // unit
    base, extra = divmod(n_units, world)
    counts, left = [], n_rays
    for r in range(world):
        c = min((base + (1 if r < extra else 0)) * unit, left)
        counts.append(c)
        left -= c
    return counts


def shard_bounds(n_rays, world, rank, unit=1024):
    counts = shard_counts(n_rays, world, unit)
    lo = sum(counts[:rank])
    return lo, lo + counts[rank]


_buffers = {}


def _buffer(tag, shape, like):
    """Grow-never, reuse-always scratch tensors: a steady-state frame loop allocates nothing for the exchange."""
    key = (tag, tuple(shape), like.device, like.dtype)
    buf = _buffers.get(key)
    if buf is None:
        buf = _buffers[key] = torch.zeros(*shape, device=like.device, dtype=like.dtype)
    return buf


def clear_buffers():
    """Drop the cached exchange buffers (they are otherwise kept for the life of the process)."""
    _buffers.clear()


def gather_tiles(tile, n_rays, world, unit=1024, group=None, out=None, reuse=False):
    """All ranks' (r_i, C) tiles -> the full (n_rays, C) frame on every rank.
    Unequal shards are padded to the largest so one all_gather_into_tensor suffices.  The padded send tile and the
    receive buffer are scratch, cached per (shape, device, dtype, group).  The RETURNED frame is a fresh tensor by
    default (callers may keep frame k while frame k+1 is rendered); pass `out=` (an (n_rays, C) tensor to fill) or
    `reuse=True` (a cached frame buffer, overwritten by the next call of the same shape: a steady-state loop that
    consumes each frame before rendering the next allocates nothing)."""
    import torch.distributed as dist
    counts = shard_counts(n_rays, world, unit)
    biggest = max(counts)
    C = tile.shape[1]
    gid = id(group) if group is not None else 0
    if tile.shape[0] != biggest or not tile.is_contiguous():
        pad = _buffer(("send", gid), (biggest, C), tile)
        pad[: tile.shape[0]].copy_(tile)
        tile = pad
    equal = all(c == biggest for c in counts)
    if out is not None:
        assert tuple(out.shape) == (n_rays, C) and out.is_contiguous(), "out must be a contiguous (n_rays, C) tensor"
        frame = out
    elif reuse:
        frame = _buffer(("frame", gid), (n_rays, C), tile)
    else:
        frame = torch.empty(n_rays, C, device=tile.device, dtype=tile.dtype)
    if equal:                                  # gather straight into the frame: no compaction pass
        dist.all_gather_into_tensor(frame, tile, group=group)
        return frame
    recv = _buffer(("recv", gid), (world * biggest, C), tile)
    dist.all_gather_into_tensor(recv, tile, group=group)
    lo = 0
    for r in range(world):
        frame[lo: lo + counts[r]].copy_(recv[r * biggest: r * biggest + counts[r]])
        lo += counts[r]
    return frame


def alter_gather_cat(outputs, key, image_sizes, group=None):
    """The reference's evaluation gather for a LIST of images per rank (LitModel.alter_gather_cat, models/interface.py:30-50;
    call sites neo360/model.py:1073-1086, vanilla_nerf/model.py:380-393): every rank holds the outputs of ITS test images
    (`outputs[i][key]`: (h w, 3) colours or (h w,) / (h w, 1) scalars); they are concatenated, all-gathered into (world, n, C),
    reordered with the reference's `permute(1, 0, 2).flatten(0, 1)` - row j of rank 0, row j of rank 1, ... - and cut into
    images of `image_sizes` (h, w).  ONE all_gather_into_tensor (RCCL on devices, gloo on CPU tensors) instead of Lightning's
    `self.all_gather`; all ranks must hold the same number of rows, as Lightning's all_gather requires.  The row interleaving
    is the reference's behaviour at world > 1 and is reproduced as is (so is its failure on 1-D per-pixel scalars at world > 1:
    pass them as (n, 1), as the reference's call sites do).  Without an initialised process group (single process) the gather
    is the identity, as Lightning's is."""
    import torch.distributed as dist
    each = torch.cat([o[key] for o in outputs]).detach()
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if world > 1:
        flat = each.contiguous()
        every = torch.empty((world,) + tuple(flat.shape), device=flat.device, dtype=flat.dtype)
        dist.all_gather_into_tensor(every.view(world * flat.shape[0], *flat.shape[1:]) if flat.dim() > 0 else every, flat, group=group)
        all_ = every
    else:
        all_ = each                              # Lightning's all_gather outside a distributed run returns its input
    if all_.dim() == 3:
        all_ = all_.permute((1, 0, 2)).flatten(0, 1)
    if all_.dim() >= 1 and all_.shape[-1] == 1:
        all_ = all_.squeeze(-1)
    ret, curr = [], 0
    for (h, w) in image_sizes:
        part = all_[curr: curr + h * w]
        if all_.dim() == 2 and all_.shape[-1] == 3:
            if part.shape[0] == 0:
                continue
            ret.append(part.reshape(h, w, 3))
        else:
            ret.append(part.reshape(h, w))
        curr += h * w
    return ret


def ray_patch_order(n_rays, width, first_ray=0, pw_log2=1, ph_log2=1):
    """Host mirror of csrc/tp_common.h:patch_point at ray granularity (documentation + CPU test; the device function is what runs):
    the order in which the NeO-360 evaluators visit the rays of a launch that carries the pixel-grid hint (neo_ctx_set_ray_grid).
    Returns a LongTensor `order` with order[k] = the ray (0 .. n_rays-1) visited k-th: inside every WHOLE band of 2^ph image rows
    that lies inside [first_ray, first_ray + n_rays) the rays go patch by patch (2^pw x 2^ph pixels, row-major inside a patch,
    patches left to right); rays outside whole bands keep their place.  A permutation for every (n_rays, first_ray)."""
    import torch
    k = torch.arange(n_rays, dtype=torch.long)
    if width <= 0 or width % (1 << pw_log2) != 0:
        return k
    band = width << ph_log2
    G = first_ray + k
    b = G // band
    whole = (b * band >= first_ray) & ((b + 1) * band <= first_ray + n_rays)
    kk = G - b * band
    r = kk & ((1 << (pw_log2 + ph_log2)) - 1)
    Gt = b * band + (r >> pw_log2) * width + ((kk >> (pw_log2 + ph_log2)) << pw_log2) + (r & ((1 << pw_log2) - 1))
    return torch.where(whole, Gt - first_ray, k)
