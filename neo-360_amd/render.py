"""Whole-frame rendering: the counterpart of the reference's Lit*.render_rays_test
chunk loops (vanilla_nerf/model.py:336-363, neo360/model.py:861-907) without the
Python loop — one library call per frame (or per rank's shard of it), with the
reference chunk size passed down so NeO-360's chunk-dependent view-direction tiling
is reproduced — plus the single-frame replacement of the reference's only collective
(LitModel.alter_gather_cat, models/interface.py:30-50) and its PSNR (:53-61).
"""
import math
import warnings

import torch

from . import _lib, models
from .parallel import gather_tiles, shard_bounds

_WHOLE = ("src_imgs", "src_poses", "src_focal", "src_c")


def _slice(batch, lo, hi):
    out = {}
    for k, v in batch.items():
        if k in _WHOLE or not isinstance(v, torch.Tensor):
            out[k] = v
        elif k == "radii" and v.dim() == 2 and v.shape[0] == 1:
            out[k] = v[:, lo:hi]
        else:
            out[k] = v[lo:hi]
    return out


def _render_once(model, batch, chunk, white_bkgd, near, far, train_frac):
    if isinstance(model, models.NeRF_TP):
        res = model(batch, False, white_bkgd, near, far, out_depth=True, chunk=chunk)
        return dict(rgb=res[1][0], depth=res[1][5], fg_rgb=res[1][1], bg_rgb=res[1][2], acc=res[1][3])
    if isinstance(model, models.PixelNeRF):
        res = model(batch, False, white_bkgd, near, far, chunk=chunk)      # vanilla_nerf/model_pixel.py:356-383
        return dict(rgb=res[1][0], depth=res[1][2], acc=res[1][1])
    if isinstance(model, models.MipNeRF360):
        # mipnerf360/model.py:471-505: train_frac = global_step / max_steps of the trainer; near/far as given
        rend, hist = model(batch, train_frac, False, False, near, far)
        w = hist[-1]["weights"]
        return dict(rgb=rend[-1]["rgb"], acc=w.sum(-1), depth=torch.zeros_like(w[:, 0]))   # the reference returns rgb only
    if isinstance(model, models.NeRF):
        res = model(batch, False, white_bkgd, near, far)     # chunking does not change vanilla results
        return dict(rgb=res[1][0], depth=res[1][2], acc=res[1][1])
    raise TypeError("unsupported renderer %r" % type(model))


def _render_exact(model, batch, chunk, white_bkgd, near, far, train_frac):
    prev = model.precision
    model.precision = "f32"
    try:
        out = _render_once(model, batch, chunk, white_bkgd, near, far, train_frac)
        model.check_flags()
    finally:
        model.precision = prev
    out["precision_used"] = "f32"
    return out


@torch.no_grad()
def render_rays_test(model, batch, chunk=1024, white_bkgd=False, near=0.2, far=3.0, train_frac=1.0, check=True,
                     on_range="retry_f32", image_width=None, first_ray=0):
    """Fine-level rgb / depth of every ray in `batch` (one image), as the reference's
    render_rays_test returns them: dict(rgb (R,3), depth (R,)) plus `target` /
    `instance_mask` passed through when present.  check=True waits for the frame and raises what the device-side
    assertions reported (a ray that missed the unit sphere); check=False leaves that to
    a later `model.check_flags()` (a loop over frames that never wants to block).

    on_range: what happens when the range guard of the split-fp16 arithmetic trips for this frame (an operand beyond the
    fp16 range: a trained checkpoint or an un-normalised encoder; the reference is plain fp32 and never fails,
    neo360/model.py:343-407).  "retry_f32" (default): the frame is rendered again on the exact fp32-MFMA kernels of the
    same library - bitwise the frame `model.precision = "f32"` returns - and a RuntimeWarning says so once per module
    (the flag lives ON the module); "raise": the NeoRangeError goes to the caller.  Needs check=True (the guard is read when
    the frame is complete).  When the operand that left the range is a STATIC one (packed weights, an uploaded feature map:
    `NeoRangeError.static_operand`) every later frame would trip again, so the module is LATCHED: until its parameters or
    its scene change (`model.operands_key()`), frames go straight to the exact kernels - one render per frame, not a failed
    split attempt plus a retry.  A trip on activations is per frame and is not latched.  `out["precision_used"] = "f32"`
    marks such frames; `model.last_precision_used` holds the arithmetic of the last frame either way."""
    if on_range not in ("retry_f32", "raise"):
        raise ValueError("on_range must be 'retry_f32' or 'raise', got %r" % (on_range,))
    if image_width and isinstance(model, models.NeRF_TP):
        # image_width (+ first_ray: the frame index of this batch's first ray, for a rank's shard): the rays are the row-major
        # pixels of an image - the evaluators then walk them in 8 x 8 pixel patches (neo_ctx_set_ray_grid: L2 locality in both
        # image directions, bitwise the same frame)
        prev_grid = getattr(model, "ray_grid", None)
        model.ray_grid = (int(image_width), int(first_ray))
        try:
            return render_rays_test(model, batch, chunk, white_bkgd, near, far, train_frac, check, on_range)
        finally:
            model.ray_grid = prev_grid
    latch = getattr(model, "_range_latch", None)
    if latch is not None and on_range == "retry_f32" and check and (model.precision or model.default_precision) != "f32":
        if latch == model.operands_key():
            out = _render_exact(model, batch, chunk, white_bkgd, near, far, train_frac)
            model.last_precision_used = "f32"
            for k in ("target", "instance_mask"):
                if k in batch:
                    out[k] = batch[k]
            return out
        model._range_latch = None          # weights or scene changed: the split arithmetic gets another chance
    out = _render_once(model, batch, chunk, white_bkgd, near, far, train_frac)
    model.last_precision_used = model.precision or model.default_precision
    if check:
        try:
            model.check_flags()      # the deferred reads of the assertion word: raise before the frame is handed out
        except _lib.NeoRangeError as err:
            if on_range != "retry_f32":
                raise
            if not getattr(model, "_warned_range_downgrade", False):
                model._warned_range_downgrade = True
                warnings.warn("%s: an operand left the fp16 range of the split arithmetic (precision 'f16x3'); this frame "
                              "(and any later one that trips the guard) is re-rendered on the exact fp32 kernels "
                              "(~3-6x slower). Set model.precision = 'f32' to skip the failed attempt."
                              % type(model).__name__, RuntimeWarning, stacklevel=2)
            if getattr(err, "static_operand", False):
                model._range_latch = model.operands_key()
            out = _render_exact(model, batch, chunk, white_bkgd, near, far, train_frac)
            model.last_precision_used = "f32"
    for k in ("target", "instance_mask"):
        if k in batch:
            out[k] = batch[k]
    return out


@torch.no_grad()
def render_frame_sharded(model, batch, world, rank, chunk=1024, white_bkgd=False, near=0.2, far=3.0, group=None,
                         gather=True, train_frac=1.0, n_rays=None, out=None, reuse=False, check=True, always_gather=False,
                         info=None, image_width=None):
    """This rank renders its contiguous range of whole chunks; `gather=True` reassembles
    the full (R,5) = (rgb, depth, acc) frame on every rank with one all-gather.
    `batch` holds the whole frame's rays, or - with n_rays = R given - only this rank's shard
    (rays [shard_bounds(R, world, rank)), e.g. from ops.get_ray_directions_and_rays(ray_range=...)).
    The gathered frame is a fresh tensor unless `out=` / `reuse=True` are given (parallel.gather_tiles).
    always_gather: run the collective at world == 1 too (a one-rank process group: the RCCL call path of the N-GPU job
    on a one-GPU box; bench.py under `torchrun --nproc-per-node 1`).
    info: an optional dict that receives `precision_used` = the arithmetic THIS rank's shard was rendered in ("f16x3" /
    "f32": a range-guard retry, see render_rays_test) and, when the frame is gathered over more than one rank,
    `precision_by_rank` (one extra 4-byte all-gather, only when `info` is given): a frame whose ranks mixed arithmetics
    says so instead of hiding it in the gathered tile."""
    if n_rays is None:
        R = batch["rays_o"].shape[0]
        lo, hi = shard_bounds(R, world, rank, unit=chunk)
        mine = _slice(batch, lo, hi)
    else:
        R = int(n_rays)
        lo, hi = shard_bounds(R, world, rank, unit=chunk)
        assert batch["rays_o"].shape[0] == hi - lo, "batch must hold exactly this rank's shard"
        mine = batch
    part = render_rays_test(model, mine, chunk, white_bkgd, near, far, train_frac, check=check, image_width=image_width, first_ray=lo)
    tile = torch.cat([part["rgb"], part["depth"][:, None], part["acc"][:, None]], dim=1)
    used = part.get("precision_used") or getattr(model, "last_precision_used", None) or model.precision or model.default_precision
    if info is not None:
        info["precision_used"] = used
    if not gather or (world == 1 and not always_gather):
        return tile
    if info is not None and world > 1:
        import torch.distributed as dist
        mine_p = torch.tensor([1 if used == "f32" else 0], dtype=torch.int32, device=tile.device)
        every = torch.empty(world, dtype=torch.int32, device=tile.device)
        dist.all_gather_into_tensor(every, mine_p, group=group)
        info["precision_by_rank"] = ["f32" if int(x) else "f16x3" for x in every.tolist()]
    return gather_tiles(tile, R, world, unit=chunk, group=group, out=out, reuse=reuse)


def psnr(pred, gt):
    """-10 ln(mse)/ln 10 on images clipped to [0,1] (models/interface.py:53-61)."""
    mse = torch.mean((torch.clip(pred, 0, 1) - torch.clip(gt, 0, 1)) ** 2)
    return float("inf") if float(mse) == 0.0 else float(-10.0 * torch.log(mse) / math.log(10))
