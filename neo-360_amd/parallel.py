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


def gather_tiles(tile, n_rays, world, unit=1024, group=None):
    """All ranks' (r_i, C) tiles -> the full (n_rays, C) frame on every rank.
    Unequal shards are padded to the largest so one all_gather_into_tensor suffices.  The padded send tile, the
    receive buffer and (for unequal shards) the compacted frame are cached per shape: the returned tensor is
    overwritten by the next call with the same shapes."""
    import torch.distributed as dist
    counts = shard_counts(n_rays, world, unit)
    biggest = max(counts)
    C = tile.shape[1]
    if tile.shape[0] != biggest or not tile.is_contiguous():
        pad = _buffer("send", (biggest, C), tile)
        pad[: tile.shape[0]].copy_(tile)
        tile = pad
    out = _buffer("recv", (world * biggest, C), tile)
    dist.all_gather_into_tensor(out, tile, group=group)
    if all(c == biggest for c in counts):
        return out
    frame = _buffer("frame", (n_rays, C), tile)
    lo = 0
    for r in range(world):
        frame[lo: lo + counts[r]].copy_(out[r * biggest: r * biggest + counts[r]])
        lo += counts[r]
    return frame
