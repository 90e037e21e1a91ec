"""Drop-in renderer modules: same constructor arguments, `forward` signatures,
return tuples and `state_dict` keys as the reference's renderer nn.Modules, so
they slot in under the reference's Lit* systems / run.py unchanged (SURVEY.md
§8b).  The parameters are ordinary nn.Linear containers; all arithmetic happens
in the HIP library.

  NeRF      <-> models/vanilla_nerf/model.py:128-216
  NeRF_TP   <-> models/neo360/model.py:162-581 (decoder half; scene features
                come from `set_scene`, or from an attached encoder module)

Inference path only (`randomized=False`, no autograd), as SURVEY.md §8b scopes it.
"""
import ctypes
import math
import os
import warnings
import weakref

import torch
import torch.nn as nn

from . import _lib
from .context import CallOverlap, f32, new_context, ptr


def _xavier_linear(n_in, n_out, xavier=True):
    layer = nn.Linear(n_in, n_out)
    if xavier:
        nn.init.xavier_uniform_(layer.weight)
    return layer


def _fingerprint(tensors):
    return tuple((t.data_ptr(), t._version, t.device) for t in tensors)


class _SceneKey:
    """Identity of the inputs an attached encoder was last run on.  Holds STRONG references to the source tensors
    (so the allocator cannot hand their addresses to a later batch) and compares them with `is` + `_version`,
    plus a fingerprint of the encoder's parameters / buffers and its training flag: a new `src_imgs` tensor, an
    in-place edit, `load_state_dict`, an optimizer step or `.train()` all force a re-encode."""

    def __init__(self, tensors, encoder):
        self.tensors = tuple(tensors)
        self.versions = tuple(t._version for t in self.tensors)
        self.enc = self._enc_fp(encoder)

    @staticmethod
    def _enc_fp(encoder):
        if not isinstance(encoder, nn.Module):
            return None
        items = list(encoder.parameters()) + list(encoder.buffers())
        return (encoder.training, tuple((id(t), t.data_ptr(), t._version) for t in items))

    def matches(self, tensors, encoder):
        tensors = tuple(tensors)
        return (len(tensors) == len(self.tensors) and all(a is b for a, b in zip(tensors, self.tensors))
                and tuple(t._version for t in tensors) == self.versions and self._enc_fp(encoder) == self.enc)


def _ptr_table(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


class NeRFMLP(nn.Module):
    """Parameter container with the layout of vanilla_nerf/model.py:44-98:
    pts_linears.0..7 (256 wide, skip concat feeds index 5), views_linear.0,
    bottleneck_layer, density_layer, rgb_layer.  Initialisers as the reference
    (xavier_uniform_ on every weight except views_linear.0, default biases)."""

    def __init__(self, min_deg_point=0, max_deg_point=10, deg_view=4, netdepth=8, netwidth=256,
                 netdepth_condition=1, netwidth_condition=128, skip_layer=4, input_ch=3, input_ch_view=3,
                 num_rgb_channels=3, num_density_channels=1):
        super().__init__()
        if (netdepth, netwidth, netdepth_condition, netwidth_condition, skip_layer, input_ch, input_ch_view,
                num_rgb_channels, num_density_channels, min_deg_point, max_deg_point, deg_view) != (
                8, 256, 1, 128, 4, 3, 3, 3, 1, 0, 10, 4):
            raise NotImplementedError("the HIP kernel is specialised for the reference's default NeRFMLP shape")
        pos = ((max_deg_point - min_deg_point) * 2 + 1) * input_ch
        view = (deg_view * 2 + 1) * input_ch_view
        layers = [_xavier_linear(pos, netwidth)]
        for idx in range(netdepth - 1):
            layers.append(_xavier_linear(netwidth + pos if (idx % skip_layer == 0 and idx > 0) else netwidth, netwidth))
        self.pts_linears = nn.ModuleList(layers)
        self.views_linear = nn.ModuleList([_xavier_linear(netwidth + view, netwidth_condition, xavier=False)])
        self.bottleneck_layer = _xavier_linear(netwidth, netwidth)
        self.density_layer = _xavier_linear(netwidth, num_density_channels)
        self.rgb_layer = _xavier_linear(netwidth_condition, num_rgb_channels)

    def ordered_layers(self):
        """Upload order fixed by include/neo360_hip.h (neo_vanilla_upload_mlp)."""
        return list(self.pts_linears) + [self.views_linear[0], self.bottleneck_layer, self.density_layer, self.rgb_layer]


class _HipModule(nn.Module):
    """Shared plumbing: one private library context per module and device,
    parameters re-packed on the device whenever they change."""

    def __init__(self):
        super().__init__()
        self._ctx_cache = {}

    # MLP GEMM arithmetic: "f16x3" = fp16 matrix cores on hi/lo-split fp32 operands (fp32-equivalent,
    # ~3x faster), "f32" = exact fp32 MFMA.  None -> $NEO360_PRECISION or the class default.
    precision = None
    default_precision = "f16x3"
    # The device assertion word (bit 0: the unit-sphere assertion of NeRF_TP, bit 1: the range guard of the split-fp16
    # arithmetic) is read WITHOUT synchronising by default (no sync inside the call):
    #   "deferred"  (default) every call enqueues a read-and-clear of the word behind its kernels; reads that have
    #               completed are looked at when the NEXT call on this module starts, and all of them in
    #               `check_flags()` (which waits) - render.render_rays_test calls it before returning a frame, and a
    #               caller running its own chunk loop calls `model.check_flags()` once before it consumes the results;
    #   "immediate" / True   one stream synchronisation + read after every call: the exception is raised by the very
    #               call that tripped it, exactly like the reference's assert (300 syncs per frame under a chunk loop);
    #   "never" / False   no reads at all; `check_flags()` reads the word on demand.
    poll_flags = "deferred"

    def _flag_mode(self):
        m = self.poll_flags
        if m is True or m == "immediate":
            return "immediate"
        if m is False or m is None or m == "never":
            return "never"
        if m == "deferred":
            return "deferred"
        raise ValueError("poll_flags must be 'deferred', 'immediate' (True) or 'never' (False), got %r" % (m,))

    def check_flags(self):
        """Wait for every outstanding read of the assertion word and raise what it reports (AssertionError for a ray
        that missed the unit sphere, NeoError for the split-fp16 range guard)."""
        for ctx in self._ctx_cache.values():
            flags = ctx.take_flags(wait=True)
            if self._flag_mode() == "never":
                flags |= ctx.poll_flags()
            self._raise_flags(flags)

    def _before_call(self, ctx):
        if self._flag_mode() == "deferred":
            self._raise_flags(ctx.take_flags(wait=False), late=True)      # completed reads of earlier calls; never blocks

    def _after_call(self, ctx):
        mode = self._flag_mode()
        if mode == "immediate":
            self._raise_flags(ctx.poll_flags())
        elif mode == "deferred":
            ctx.post_flags()

    def _context(self, device):
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
        ctx = self._ctx_cache.get(key)
        if ctx is None:
            ctx = self._ctx_cache[key] = new_context(device)     # freed by Context's finalizer with the module
        want = self.precision or os.environ.get("NEO360_PRECISION", self.default_precision)
        if getattr(ctx, "_precision", None) != want:
            ctx.set_precision(want)
            ctx._precision = want
        return ctx

    def close(self):
        """Release the library contexts (packed weights, scene features, workspaces) now.  Deferred reads of the
        assertion word that nobody looked at (`poll_flags='deferred'` and no later call / `check_flags()`) are taken
        here; a tripped assertion or range guard becomes a RuntimeWarning instead of vanishing."""
        for ctx in self._ctx_cache.values():
            if ctx.handle and self._flag_mode() == "deferred":
                try:
                    flags = ctx.take_flags(wait=True)
                except Exception:
                    flags = 0
                if flags:
                    warnings.warn("%s.close(): unread device assertions of earlier calls (flags=%d: %s); results of those "
                                  "calls were invalid - call check_flags() before consuming outputs"
                                  % (type(self).__name__, flags & 3, " + ".join(
                                      n for b, n in ((1, "ray missed the unit sphere"), (2, "split-fp16 range guard")) if flags & b)),
                                  RuntimeWarning, stacklevel=2)
            ctx.close()
        self._ctx_cache.clear()

    @staticmethod
    def _raise_flags(flags, late=False):
        """Device assertion word -> the reference's AssertionError / a NeoError for the split-fp16 range guard.
        late: the word was posted by an EARLIER call on this module (deferred mode)."""
        where = " (raised by an earlier call on this module: poll_flags='deferred')" if late else ""
        if flags & 1:
            raise AssertionError("1.0 - p_norm_sq should be greater than 0" + where)
        if flags & 8:
            raise _lib.NeoError("internal error reported by a kernel (flag bit 3: a bounded in-kernel wait ran out); results of that "
                                "call are invalid" + where)
        if flags & 2:
            err = _lib.NeoRangeError(
                "split-fp16 arithmetic (precision 'f16x3') met an operand outside the fp16 range (|x| >= 65504 or "
                "non-finite %s): results of that call are invalid; use precision 'f32'"
                % ("weights / feature maps" if flags & 4 else "activations / gathered features") + where)
            err.static_operand = bool(flags & 4)
            raise err

    @staticmethod
    def _check_mode(randomized):
        if randomized:
            raise NotImplementedError(
                "randomized=True (stratified jitter / training) is outside the accelerated inference path")

    # Which path a training-shaped call takes (NeRF.forward, NeRF_TP.forward(out_depth=False)):
    #   None (default)  automatic: the differentiable operator chain of training.py when autograd is on and anything that
    #                   could receive a gradient requires one - a parameter of this module (incl. an attached encoder's) or
    #                   a scene tensor given to set_scene; otherwise the fused no-grad kernels;
    #   True / False    always / never the operator chain, whatever autograd says.
    # The operator chain materialises per-point-view feature rows and an activation tape, re-runs an attached encoder per
    # call and synchronises once per call: a forward-only caller outside torch.no_grad() with default nn.Module parameters
    # (requires_grad=True) takes it unless it says `module.differentiable = False` (or wraps the call in no_grad).
    differentiable = None
    _range_latch = None            # render.render_rays_test: operands_key() of the (weights, scene) whose STATIC operands tripped the range guard
    last_precision_used = None     # arithmetic of the last frame render.render_rays_test produced on this module

    # Consecutive fused evaluation calls of one module overlap on the device (context.CallOverlap: two side streams x two scratch
    # lanes of the library context; the caller's stream semantics are unchanged).  What it is for: the reference's own chunk loops
    # - 300 forward calls of 1024 rays per frame (neo360/model.py:861-907, vanilla_nerf/model.py:336-363, model_pixel.py:356-383,
    # mipnerf360/model.py:471-505) - whose calls would otherwise each end on a partly filled machine.
    # False / $NEO360_OVERLAP=0: every call runs on the caller's stream, as before round 6.
    overlap_calls = os.environ.get("NEO360_OVERLAP", "1") != "0"

    def _overlap(self, dev):
        if not self.overlap_calls:
            return None
        table = self.__dict__.setdefault("_overlap_state", {})
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        ov = table.get(key)
        if ov is None:
            ov = table[key] = CallOverlap(torch.device("cuda", key))
        return ov

    def _launch_overlapped(self, ctx, dev, raw, conv, launch):
        """Run `launch()` - output allocation + ONE fused library call + the flag post - on this call's side stream and scratch
        lane (or on the caller's stream when overlap is off); `launch` returns the list of output tensors.  raw / conv: the ray
        tensors as the caller passed them / as the library reads them (fork-point reuse: CallOverlap.begin)."""
        ov = self._overlap(dev)
        if ov is None:
            return launch()
        side, lane, cur = ov.begin(raw, conv)
        ctx.set_lane(lane)
        outs = []
        try:
            with torch.cuda.stream(side):
                outs = launch()
        finally:
            ctx.set_lane(0)
            ov.end(lane, cur, outs)
        return outs

    def _scene_tensors_for_grad(self):
        return ()

    def operands_key(self):
        """Identity of the STATIC operands of the split arithmetic: every parameter (address + version) and the uploaded
        scene.  render.render_rays_test latches a module whose weights / feature maps tripped the range guard to the exact
        kernels for as long as this key stays the same."""
        return (tuple((p.data_ptr(), p._version) for p in self.parameters()), getattr(self, "_scene_serial", 0))

    def _wants_grad(self, randomized=True):
        """randomized: the call's own flag.  An AUTOMATIC switch to the operator chain on a deterministic call (the shape of a
        forward-only caller that merely forgot torch.no_grad(): nn.Module parameters default to requires_grad=True) is announced
        once per module - such a caller wants `module.differentiable = False` or no_grad; a training loop that
        really differentiates a deterministic call sets `module.differentiable = True` and never sees the message."""
        if self.differentiable is not None:
            return bool(self.differentiable)
        if not torch.is_grad_enabled():
            return False
        auto = (any(p.requires_grad for p in self.parameters())
                or any(t is not None and t.requires_grad for t in self._scene_tensors_for_grad()))
        if auto and not randomized and not getattr(self, "_warned_auto_chain", False):
            self._warned_auto_chain = True
            warnings.warn("%s: autograd is on and a parameter / scene tensor requires grad, so this deterministic forward runs on "
                          "the differentiable operator chain (activation tapes, one synchronisation per call) instead of the fused "
                          "kernels.  Forward-only callers: wrap the call in torch.no_grad() or set module.differentiable = False; "
                          "training code: set module.differentiable = True to silence this." % type(self).__name__,
                          RuntimeWarning, stacklevel=3)
        return auto

    def _auto_chain(self):
        """True when the operator chain would be taken by the automatic rule only (nobody asked for it explicitly)."""
        return self.differentiable is None


class NeRF(_HipModule):
    """Vanilla coarse+fine NeRF renderer (vanilla_nerf/model.py:128-216)."""

    def __init__(self, num_levels=2, min_deg_point=0, max_deg_point=10, deg_view=4, num_coarse_samples=64,
                 num_fine_samples=128, use_viewdirs=True, noise_std=0.0, lindisp=False):
        super().__init__()
        if num_levels != 2 or not use_viewdirs or lindisp:
            raise NotImplementedError("only the reference's default 2-level, view-dependent, linear-depth setup")
        self.num_levels = num_levels
        self.min_deg_point, self.max_deg_point, self.deg_view = min_deg_point, max_deg_point, deg_view
        self.num_coarse_samples, self.num_fine_samples = num_coarse_samples, num_fine_samples
        self.use_viewdirs, self.noise_std, self.lindisp = use_viewdirs, noise_std, lindisp
        self.coarse_mlp = NeRFMLP(min_deg_point, max_deg_point, deg_view)
        self.fine_mlp = NeRFMLP(min_deg_point, max_deg_point, deg_view)

    def _sync_weights(self, ctx):
        for slot, mlp in enumerate((self.coarse_mlp, self.fine_mlp)):
            layers = mlp.ordered_layers()
            ws = [f32(l.weight.detach(), "weight") for l in layers]
            bs = [f32(l.bias.detach(), "bias") for l in layers]
            fp = _fingerprint(ws + bs)
            if ctx.uploaded.get(("vanilla", slot)) == fp:
                continue
            _lib.check(ctx.lib.neo_vanilla_upload_mlp(ctx.handle, slot, _ptr_table(ws), _ptr_table(bs), ctx.stream()))
            ctx.uploaded[("vanilla", slot)] = fp

    def forward(self, rays, randomized, white_bkgd, near, far, seed=None):
        """Returns [(rgb (B,3), acc (B,), depth (B,))] * 2, as the reference.  randomized=True (stratified samples, random
        quantiles, `noise_std`) or a call that wants gradients (`_wants_grad`: the reference's training_step,
        vanilla_nerf/model.py:281-283) runs on the differentiable operators of training.py (nerf_render_train: NeRFMLP
        with a native backward, samplers on the counter-based generator, compositing with a native backward);
        the deterministic no-grad call is ONE fused library call."""
        if randomized or self._wants_grad(randomized):
            from . import training
            return training.nerf_render_train(self, rays, randomized, white_bkgd, near, far, seed)
        with torch.no_grad():
            return self._forward_fused(rays, white_bkgd, near, far)

    def _forward_fused(self, rays, white_bkgd, near, far):
        raw = (rays["rays_o"], rays["viewdirs"], rays["rays_d"])
        rays_o = f32(raw[0], "rays_o")
        viewdirs = f32(raw[1], "viewdirs")
        rays_d = f32(raw[2], "rays_d")
        dev = rays_o.device
        ctx = self._context(dev)
        self._sync_weights(ctx)
        self._before_call(ctx)
        B = rays_o.shape[0]
        outs = []

        def launch():
            outs.extend((torch.empty(B, 3, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev)) for _ in range(2))
            _lib.check(ctx.lib.neo_vanilla_render(
                ctx.handle, ptr(rays_o), ptr(viewdirs), ptr(rays_d), B, float(near), float(far),
                self.num_coarse_samples, self.num_fine_samples, int(bool(white_bkgd)),
                ptr(outs[0][0]), ptr(outs[0][1]), ptr(outs[0][2]), ptr(outs[1][0]), ptr(outs[1][1]), ptr(outs[1][2]),
                ctx.stream()))
            self._after_call(ctx)
            return [t for lv in outs for t in lv]
        self._launch_overlapped(ctx, dev, raw, (rays_o, viewdirs, rays_d), launch)
        return outs

    @torch.no_grad()
    def eval_mlp(self, level, rays_o, dirs, t):
        """Stage-level access for parity tests: pos_enc + MLP + activations at
        points o + t*dirs.  t (B,N) -> (B,N,4) = (rgb, sigma)."""
        rays_o, dirs, t = f32(rays_o), f32(dirs), f32(t)
        ctx = self._context(rays_o.device)
        self._sync_weights(ctx)
        B, N = t.shape
        out = torch.empty(B, N, 4, device=rays_o.device)
        _lib.check(ctx.lib.neo_vanilla_mlp(ctx.handle, level, ptr(rays_o), ptr(dirs), ptr(t), N, B, N, ptr(out),
                                           ctx.stream()))
        self._raise_flags(ctx.poll_flags())     # stage-level access: always immediate
        return out


class NeRFPPMLP(nn.Module):
    """Parameter container with the layout of neo360/model.py:37-108: pts_linears.0..3
    (128 wide; input = pos_enc + 512 local + 128 world; skip concat feeds index 3),
    views_linear.0/.1 (64), bottleneck_layer, density_layer, rgb_layer."""

    def __init__(self, min_deg_point=0, max_deg_point=10, deg_view=4, netdepth=4, netwidth=128,
                 netdepth_condition=2, netwidth_condition=64, skip_layer=2, input_ch=3, input_ch_view=3,
                 num_rgb_channels=3, num_density_channels=1, local_latent_size=512, world_latent_size=128,
                 combine_layer=3, combine_type="average", out_nocs=False, num_src_views=3):
        super().__init__()
        if (netdepth, netwidth, netdepth_condition, netwidth_condition, skip_layer, input_ch_view, num_rgb_channels,
                num_density_channels, local_latent_size, world_latent_size, combine_layer, combine_type, out_nocs,
                min_deg_point, max_deg_point, deg_view) != (4, 128, 2, 64, 2, 3, 3, 1, 512, 128, 3, "average", False,
                                                            0, 10, 4) or input_ch not in (3, 4):
            raise NotImplementedError("the HIP kernel is specialised for the reference's default NeRFPPMLP shape")
        self.input_ch = input_ch
        self.num_src_views = num_src_views
        pos = ((max_deg_point - min_deg_point) * 2 + 1) * input_ch + local_latent_size + world_latent_size
        view = (deg_view * 2 + 1) * input_ch_view
        layers = [_xavier_linear(pos, netwidth)]
        for idx in range(netdepth - 1):
            layers.append(_xavier_linear(netwidth + pos if (idx % skip_layer == 0 and idx > 0) else netwidth, netwidth))
        self.pts_linears = nn.ModuleList(layers)
        self.views_linear = nn.ModuleList([_xavier_linear(netwidth + view, netwidth_condition, xavier=False),
                                           _xavier_linear(netwidth_condition, netwidth_condition)])
        self.bottleneck_layer = _xavier_linear(netwidth, netwidth)
        self.density_layer = _xavier_linear(netwidth, num_density_channels)
        self.rgb_layer = _xavier_linear(netwidth_condition, num_rgb_channels)

    def ordered_layers(self):
        """Upload order fixed by include/neo360_hip.h (neo_tp_upload_mlp)."""
        return list(self.pts_linears) + [self.views_linear[0], self.views_linear[1], self.bottleneck_layer,
                                         self.density_layer, self.rgb_layer]


class NeRF_TP(_HipModule):
    """NeO-360 decoder renderer (neo360/model.py:162-581).

    `forward(rays, randomized, white_bkgd, near, far, out_depth=False)` keeps the
    reference's signature and return tuples.  The scene features the reference's
    GridEncoder produces (three tri-planes + the pixel-aligned latent) are outside the
    accelerated path: provide them once per scene with `set_scene(...)`, or attach any
    `encoder` module with the reference's interface (callable -> three planes, and
    `.spatial_encoder.latent`); an attached encoder is run once per distinct `src_imgs`
    tensor instead of once per chunk (results are identical in eval mode).
    """

    def __init__(self, num_levels=2, min_deg_point=0, max_deg_point=10, deg_view=4, num_coarse_samples=128,
                 num_fine_samples=256, use_viewdirs=True, num_src_views=3, density_noise=0.0, lindisp=False,
                 xyz_min=None, xyz_max=None, is_optimize=False, encoder_type="resnet", feats_c_size=0, attn=False,
                 input_ch_view=3, use_same_stride=False, encoder=None):
        super().__init__()
        if num_levels != 2 or not use_viewdirs or lindisp:
            raise NotImplementedError("only the reference's default 2-level, view-dependent, linear-depth setup")
        self.num_levels, self.num_src_views = num_levels, num_src_views
        self.min_deg_point, self.max_deg_point, self.deg_view = min_deg_point, max_deg_point, deg_view
        self.num_coarse_samples, self.num_fine_samples = num_coarse_samples, num_fine_samples
        self.density_noise, self.lindisp, self.is_optimize = density_noise, lindisp, is_optimize
        if encoder is not None:
            self.encoder = encoder
        self.fg_coarse_mlp = NeRFPPMLP(min_deg_point, max_deg_point, deg_view, num_src_views=num_src_views)
        self.fg_fine_mlp = NeRFPPMLP(min_deg_point, max_deg_point, deg_view, num_src_views=num_src_views)
        self.bg_coarse_mlp = NeRFPPMLP(min_deg_point, max_deg_point, deg_view, num_src_views=num_src_views, input_ch=4)
        self.bg_fine_mlp = NeRFPPMLP(min_deg_point, max_deg_point, deg_view, num_src_views=num_src_views, input_ch=4)
        self.chunk = 1024            # rays per reference forward call (opt.py:195-200)
        self._scene_key = None
        self._scene_ctx = None
        # split path: gather the latent pre-projected through each MLP's first-layer weights (256 instead of 512
        # channels per tap, 51 % fewer MACs per point-view; see csrc/mlp_tp_hp.hip).  False: the reference's order.
        # 2: the tri-planes are pre-projected through the world columns as well (csrc/mlp_tp_hpp.hip): no world GEMM stage.
        # 3 (default): planes projected for the two outside-sphere MLPs only (measured best: profiles/r04_tp_hp_experiments.log).
        self.preproject = {"0": False, "1": True, "2": 2, "3": 3}.get(os.environ.get("NEO360_TP_PREPROJECT", "3"), 3)

    def _context(self, device):
        ctx = super()._context(device)
        mode = int(self.preproject) if (self.preproject in (2, 3) and self.preproject is not True) else int(bool(self.preproject))
        if getattr(ctx, "_preproject", None) != mode:
            _lib.check(ctx.lib.neo_tp_set_preproject(ctx.handle, mode))
            ctx._preproject = mode
        return ctx

    def _mlps(self):
        return (self.fg_coarse_mlp, self.fg_fine_mlp, self.bg_coarse_mlp, self.bg_fine_mlp)

    def _sync_weights(self, ctx):
        for slot, mlp in enumerate(self._mlps()):
            layers = mlp.ordered_layers()
            ws = [f32(l.weight.detach(), "weight") for l in layers]
            bs = [f32(l.bias.detach(), "bias") for l in layers]
            fp = _fingerprint(ws + bs)
            if ctx.uploaded.get(("tp", slot)) == fp:
                continue
            _lib.check(ctx.lib.neo_tp_upload_mlp(ctx.handle, slot, mlp.input_ch, _ptr_table(ws), _ptr_table(bs),
                                                 ctx.stream()))
            ctx.uploaded[("tp", slot)] = fp

    @torch.no_grad()
    def set_scene(self, plane_xz, plane_xy, plane_yz, latent, image_wh, preproject=None, _source=None):
        """Scene features in the reference's layout: planes (NV,128,Hp,Wp), latent
        (NV,512,Hf,Wf), image_wh = (W,H) of the source images the latent was encoded from
        (neo360/model.py:267-269).  Re-laid out channels-last on the device, once.
        preproject (None = keep `self.preproject`): see that attribute."""
        src = _source if _source is not None else (plane_xz, plane_xy, plane_yz, latent)
        # the device copy is about to change: until the upload below has succeeded NO tensor set matches it
        # (a failing upload must not leave a fingerprint that names the new tensors next to the old copy)
        self._scene_src = None
        self._scene_ctx = None
        self._scene_key = None        # whoever uploads through an attached encoder records its own key AFTER this call (_ensure_scene);
                                      # an upload from anywhere else (a training step's gather_features) must not leave an eval key behind
        if preproject is not None:
            self.preproject = int(preproject) if (preproject in (2, 3) and preproject is not True) else bool(preproject)
        planes = [f32(p, "plane") for p in (plane_xz, plane_xy, plane_yz)]
        latent = f32(latent, "latent")
        ctx = self._context(latent.device)
        NV, Cw, Hp, Wp = planes[0].shape
        _, Cl, Hf, Wf = latent.shape
        _lib.check(ctx.lib.neo_tp_set_scene(ctx.handle, ptr(planes[0]), ptr(planes[1]), ptr(planes[2]), NV, Cw, Hp, Wp,
                                            ptr(latent), Cl, Hf, Wf, float(image_wh[0]), float(image_wh[1]),
                                            ctx.stream()))
        torch.cuda.current_stream(latent.device).synchronize()   # inputs may be freed by the caller
        self._scene_ctx = ctx
        self._scene_serial = getattr(self, "_scene_serial", 0) + 1
        self._scene_wh = (float(image_wh[0]), float(image_wh[1]))
        # identity (weak references, compared with `is`) + versions: a (data_ptr, version, shape) fingerprint alone
        # matches a NEW tensor the caching allocator placed at a freed map's address (a training loop's fresh encoder
        # output at version 0 and the same shape) and the previous step's device copy would be gathered silently.
        # Weak, not strong: an inference caller's 0.5 GB of NCHW maps must stay freeable; a dead reference matches nothing.
        self._scene_src = (tuple(weakref.ref(t) for t in src), tuple(t._version for t in src))

    def scene_matches(self, maps):
        """True when the device-side scene is a copy of exactly these four tensor OBJECTS at their current version."""
        held = getattr(self, "_scene_src", None)
        maps = tuple(maps)
        return (self._scene_ctx is not None and held is not None and len(maps) == len(held[0])
                and all(a is r() for a, r in zip(maps, held[0])) and tuple(t._version for t in maps) == held[1])

    def scene_image_wh(self, rays=None):
        """(W,H) of the source images: from the last set_scene, else from rays["src_imgs"]."""
        wh = getattr(self, "_scene_wh", None)
        if wh is None:
            if rays is None or "src_imgs" not in rays:
                raise _lib.NeoError("image size unknown: call set_scene(...) first or pass rays['src_imgs']")
            wh = (float(rays["src_imgs"].shape[-1]), float(rays["src_imgs"].shape[-2]))
        return wh

    def _ensure_scene(self, rays, dev):
        enc = getattr(self, "encoder", None)
        if enc is None:
            if self._scene_ctx is None:
                raise _lib.NeoError("no scene features: call set_scene(...) or attach an encoder module")
            return
        src = rays["src_imgs"]
        inputs = (src, rays["src_poses"], rays["src_focal"], rays["src_c"])
        if self._scene_key is not None and self._scene_ctx is not None and self._scene_key.matches(inputs, enc):
            return
        planes = enc(*inputs)
        self.set_scene(planes[0], planes[1], planes[2], enc.spatial_encoder.latent,
                       (src.shape[-1], src.shape[-2]))
        self._scene_key = _SceneKey(inputs, enc)
        self.encoder_runs = getattr(self, "encoder_runs", 0) + 1

    def _camera_args(self, rays):
        """Host copies of the source cameras (kernel arguments).  Cached on the identity + version of the three
        tensors: the reference's chunk loop passes the same src_* tensors with every chunk, so only the first call of
        a frame reads them back from the device."""
        src = (rays["src_poses"], rays["src_focal"], rays["src_c"])
        cache = getattr(self, "_cam_cache", None)
        if cache is not None and all(a is b for a, b in zip(src, cache[0])) and tuple(t._version for t in src) == cache[1]:
            return cache[2]
        poses = src[0].detach().float().cpu().contiguous()
        NV = poses.shape[0]
        host_poses = (ctypes.c_float * (16 * NV))(*poses.reshape(-1).tolist())
        focal = float(src[1][0])                            # view 0's intrinsics for every view (model.py:242-244)
        cx, cy = (float(x) for x in src[2][0])
        args = (host_poses, NV, focal, cx, cy)
        self._cam_cache = (src, tuple(t._version for t in src), args)
        return args

    @torch.no_grad()
    def eval_mlp(self, slot, rays, tvals, far=None, chunk=None):
        """Stage-level access for parity tests: feature lookups + pos_enc + NeRFPPMLP +
        activations of one region at given sample positions.  slot 0/1 = fg coarse/fine
        (tvals = t), 2/3 = bg coarse/fine (tvals = descending inverse radius, needs far).
        Returns (B,N,4) = (rgb, sigma)."""
        rays_o, rays_d, viewdirs = f32(rays["rays_o"]), f32(rays["rays_d"]), f32(rays["viewdirs"])
        tvals = f32(tvals, "tvals")
        far = f32(far, "far") if far is not None else None
        ctx = self._context(rays_o.device)
        self._ensure_scene(rays, rays_o.device)
        self._sync_weights(ctx)
        B, N = tvals.shape
        host_poses, NV, focal, cx, cy = self._camera_args(rays)
        out = torch.empty(B, N, 4, device=rays_o.device)
        _lib.check(ctx.lib.neo_tp_mlp(ctx.handle, slot, ptr(rays_o), ptr(rays_d), ptr(viewdirs), ptr(tvals), ptr(far),
                                      B, N, int(chunk or max(B, 1)), host_poses, NV, focal, cx, cy, ptr(out),
                                      ctx.stream()))
        self._raise_flags(ctx.poll_flags())       # also clears the word: a miss here must not fail a later forward()
        return out

    @torch.no_grad()
    def sample_positions(self, rays, chunk=None):
        """The sample positions of the deterministic render of these rays, per level: [(fg_t (B,N), bg_s (B,N))] with N = 129,
        385 at the reference's counts - the rows the fused call's evaluators run at (level 1: sort(level-0 positions + the
        inverse-CDF samples of the level-0 weights), neo360/helper.py:218-249).  The deterministic training-shaped call and the
        evaluation call launch the same kernels in the same order, so these are bitwise the positions behind
        `forward(..., out_depth=True)`.  For parity work: an oracle evaluated AT these positions is comparable ray by ray, with
        no allowance for the ill-conditioning of the resampling itself (tests/test_gpu_fullsize.py)."""
        levels = self._forward_train(rays, False, False, chunk, None, _raw=True)
        return [(t["fg_t"], t["bg_t"]) for t in levels]

    def _forward_train(self, rays, randomized, white_bkgd, chunk=None, seed=None, _raw=False):
        """out_depth=False: per level (comp_rgb, fg_weights, bg_weights, fg_sdist, bg_sdist, bg_acc)
        (neo360/model.py:531-579), forward values only.  randomized=True draws the stratified level-0 jitter and the
        level-1 quantiles from the library's counter-based generator (`seed`, default: one fresh seed per call from
        torch's CPU generator, so torch.manual_seed makes runs repeatable)."""
        if self.density_noise != 0.0 and randomized:
            raise _lib.NeoError("density_noise is applied by the operator chain (training.tp_render_train), not by the fused call")
        rays_o, rays_d, viewdirs = f32(rays["rays_o"], "rays_o"), f32(rays["rays_d"], "rays_d"), f32(rays["viewdirs"], "viewdirs")
        dev = rays_o.device
        ctx = self._context(dev)
        self._ensure_scene(rays, dev)
        if self._scene_ctx is not ctx:
            raise _lib.NeoError("scene features were uploaded on a different device")
        self._sync_weights(ctx)
        self._before_call(ctx)
        B = rays_o.shape[0]
        host_poses, NV, focal, cx, cy = self._camera_args(rays)
        if randomized:
            seed = int(seed) if seed is not None else int(torch.randint(1, 2 ** 62, (1,)).item())
            seed = seed or 1
        else:
            seed = 0
        n0, n1 = self.num_coarse_samples + 1, self.num_coarse_samples + 1 + self.num_fine_samples
        levels, structs = [], []
        for n in (n0, n1):
            t = dict(rgb=torch.empty(B, 3, device=dev), fg_w=torch.empty(B, n, device=dev), bg_w=torch.empty(B, n, device=dev),
                     fg_t=torch.empty(B, n, device=dev), bg_t=torch.empty(B, n, device=dev), bg_acc=torch.empty(B, device=dev))
            levels.append(t)
            structs.append(_lib.TpTrainOut(t["rgb"].data_ptr(), t["fg_w"].data_ptr(), t["bg_w"].data_ptr(), t["fg_t"].data_ptr(),
                                           t["bg_t"].data_ptr(), t["bg_acc"].data_ptr(), None, None))
        _lib.check(ctx.lib.neo_tp_render_train(
            ctx.handle, ptr(rays_o), ptr(rays_d), ptr(viewdirs), B, int(chunk or max(B, 1)), host_poses, NV, focal, cx, cy,
            self.num_coarse_samples, self.num_fine_samples, int(bool(white_bkgd)), seed,
            ctypes.byref(structs[0]), ctypes.byref(structs[1]), ctx.stream()))
        self._after_call(ctx)
        if _raw:
            return levels
        out = []
        for t in levels:
            fg_t, bg_t = t["fg_t"], t["bg_t"]
            fg_sd = 0.5 * (fg_t[..., 1:] + fg_t[..., :-1])                                   # model.py:564-571
            fg_sd = torch.cat([fg_sd, (fg_sd[:, -1] + (fg_sd[:, -1] - fg_sd[:, -2])).unsqueeze(-1)], dim=-1)
            bg_sd = torch.cat([0.5 * (bg_t[..., 1:] + bg_t[..., :-1]), bg_t[..., -1:]], dim=-1)
            out.append((t["rgb"], t["fg_w"], t["bg_w"], fg_sd, bg_sd, t["bg_acc"]))
        return out

    def _maps_for_grad(self, rays):
        """The four scene tensors a differentiable call gathers from (and sends gradients to): an attached encoder is run
        WITH autograd on this batch, as the reference's forward does (neo360/model.py:281-300); otherwise the very tensors
        given to `set_scene`, if the caller still holds them."""
        enc = getattr(self, "encoder", None)
        if enc is not None:
            src = rays["src_imgs"]
            planes = enc(src, rays["src_poses"], rays["src_focal"], rays["src_c"])
            self._scene_wh = (float(src.shape[-1]), float(src.shape[-2]))
            return planes[0], planes[1], planes[2], enc.spatial_encoder.latent
        held = getattr(self, "_scene_src", None)
        maps = tuple(r() for r in held[0]) if held is not None else ()
        if len(maps) != 4 or any(m is None for m in maps):
            raise _lib.NeoError("a differentiable forward needs the scene tensors: attach an encoder module, or keep the four "
                                "tensors passed to set_scene(...) alive (the module holds only weak references to them)")
        return maps

    def _scene_tensors_for_grad(self):
        held = getattr(self, "_scene_src", None)
        return tuple(r() for r in held[0]) if held is not None else ()

    def forward(self, rays, randomized, white_bkgd, near, far, out_depth=False, chunk=None, seed=None):
        """out_depth=True (evaluation, neo360/model.py:521-527): per level (comp_rgb, fg_rgb, bg_rgb, fg_acc, bg_lambda,
        comp_depth), randomized=False.  out_depth=False (the training call, :531-579): per level (comp_rgb, fg_weights,
        bg_weights, fg_sdist, bg_sdist, bg_acc), randomized as asked.  `near`/`far` are ignored
        exactly as in the reference (:277-278).  All rays of the call form ONE reference chunk unless `chunk` is given
        (whole-frame rendering, see render.py).
        The training call is DIFFERENTIABLE when `_wants_grad()` says so (`self.differentiable`; default: autograd on and a
        parameter, an attached encoder's parameter or a set_scene tensor requires grad - the reference's training_step,
        model.py:697-820): it then runs on the operators of training.py (lookups, NeRFPPMLP with a native backward,
        compositing) instead of the fused no-grad kernels; same return tuple, same samples for one seed at any `chunk`."""
        if not out_depth:
            # density_noise (model.py:381-384) exists on the operator chain only: a randomized call with it takes that path
            if self._wants_grad(randomized) or (randomized and self.density_noise != 0.0):
                from . import training
                maps = None
                try:
                    maps = self._maps_for_grad(rays)
                except _lib.NeoError:
                    # the scene tensors are gone (weak references): only an EXPLICIT request for gradients makes that an error; a
                    # deterministic call that got here by the automatic rule falls back to the fused kernels
                    if randomized or not self._auto_chain():
                        raise
                if maps is not None:
                    return training.tp_render_train(self, rays, randomized, white_bkgd, maps, chunk, seed)
            with torch.no_grad():
                return self._forward_train(rays, randomized, white_bkgd, chunk, seed)
        with torch.no_grad():
            return self._forward_eval(rays, randomized, white_bkgd, chunk)

    def _forward_eval(self, rays, randomized, white_bkgd, chunk=None):
        self._check_mode(randomized)
        raw = (rays["rays_o"], rays["rays_d"], rays["viewdirs"])
        rays_o = f32(raw[0], "rays_o")
        rays_d = f32(raw[1], "rays_d")
        viewdirs = f32(raw[2], "viewdirs")
        dev = rays_o.device
        ctx = self._context(dev)
        self._ensure_scene(rays, dev)
        if self._scene_ctx is not ctx:
            raise _lib.NeoError("scene features were uploaded on a different device")
        self._sync_weights(ctx)
        self._before_call(ctx)
        B = rays_o.shape[0]
        host_poses, NV, focal, cx, cy = self._camera_args(rays)
        # pixel-grid hint (render.render_rays_test / render_frame_sharded set `ray_grid` around a frame): a scheduling hint for
        # the evaluators' tile order, bitwise-neutral; only meaningful for a call that spans whole bands of 8 image rows
        grid = getattr(self, "ray_grid", None)
        ctx.set_ray_grid(*(grid if grid and grid[0] % 8 == 0 and B >= 8 * grid[0] else (0, 0)))
        levels = []

        def launch():
            structs = []
            for _ in range(2):
                t = dict(rgb=torch.empty(B, 3, device=dev), fg_rgb=torch.empty(B, 3, device=dev),
                         bg_rgb=torch.empty(B, 3, device=dev), fg_acc=torch.empty(B, device=dev),
                         bg_lambda=torch.empty(B, 1, device=dev), depth=torch.empty(B, device=dev))
                levels.append(t)
                structs.append(_lib.TpLevelOut(*(t[k].data_ptr() for k in ("rgb", "fg_rgb", "bg_rgb", "fg_acc", "bg_lambda", "depth"))))
            _lib.check(ctx.lib.neo_tp_render(
                ctx.handle, ptr(rays_o), ptr(rays_d), ptr(viewdirs), B, int(chunk or max(B, 1)), host_poses, NV, focal, cx, cy,
                self.num_coarse_samples, self.num_fine_samples, int(bool(white_bkgd)),
                ctypes.byref(structs[0]), ctypes.byref(structs[1]), ctx.stream()))
            self._after_call(ctx)
            return [v for t in levels for v in t.values()]
        try:
            self._launch_overlapped(ctx, dev, raw, (rays_o, rays_d, viewdirs), launch)
        finally:
            ctx.set_ray_grid(0)
        return [(t["rgb"], t["fg_rgb"], t["bg_rgb"], t["fg_acc"], t["bg_lambda"], t["depth"]) for t in levels]


class PixelNeRFMLP(nn.Module):
    """Parameter container with the layout of vanilla_nerf/model_pixel.py:35-94: pts_linears.0..3 (128 wide;
    input = 63-d pos_enc + 512 pixel-aligned latent; the skip never fires at depth 4), views_linear.0/.1 (128),
    bottleneck_layer, density_layer, rgb_layer."""

    def __init__(self, min_deg_point=0, max_deg_point=10, deg_view=4, netdepth=4, netwidth=128,
                 netdepth_condition=2, netwidth_condition=128, skip_layer=4, input_ch=3, input_ch_view=3,
                 num_rgb_channels=3, num_density_channels=1, latent_size=512, combine_layer=3, combine_type="average"):
        super().__init__()
        if (min_deg_point, max_deg_point, deg_view, netdepth, netwidth, netdepth_condition, netwidth_condition,
                skip_layer, input_ch, input_ch_view, num_rgb_channels, num_density_channels, latent_size,
                combine_layer, combine_type) != (0, 10, 4, 4, 128, 2, 128, 4, 3, 3, 3, 1, 512, 3, "average"):
            raise NotImplementedError("the HIP kernel is specialised for the reference's default PixelNeRF MLP shape")
        pos = ((max_deg_point - min_deg_point) * 2 + 1) * input_ch + latent_size
        view = (deg_view * 2 + 1) * input_ch_view
        self.pts_linears = nn.ModuleList([_xavier_linear(pos, netwidth)] +
                                         [_xavier_linear(netwidth, netwidth) for _ in range(netdepth - 1)])
        self.views_linear = nn.ModuleList([_xavier_linear(netwidth + view, netwidth_condition, xavier=False),
                                           _xavier_linear(netwidth_condition, netwidth_condition)])
        self.bottleneck_layer = _xavier_linear(netwidth, netwidth)
        self.density_layer = _xavier_linear(netwidth, num_density_channels)
        self.rgb_layer = _xavier_linear(netwidth_condition, num_rgb_channels)

    def ordered_layers(self):
        """Upload order fixed by include/neo360_hip.h (neo_pix_upload_mlp)."""
        return list(self.pts_linears) + [self.views_linear[0], self.views_linear[1], self.bottleneck_layer,
                                         self.density_layer, self.rgb_layer]


class PixelNeRF(_HipModule):
    """PixelNeRF baseline decoder renderer (vanilla_nerf/model_pixel.py:133-258).

    `forward(rays, randomized, white_bkgd, near, far)` keeps the reference's signature and returns
    `[(comp_rgb, acc, depth)] x 2`.  The image encoder (ResNet-34 SpatialEncoder) is outside the accelerated
    path: provide its latent once per scene with `set_scene(latent, image_wh)`, or attach an `encoder` module with
    the reference's interface (callable on `src_imgs`, then `.latent`); an attached encoder runs once per
    distinct `src_imgs` tensor instead of once per chunk.  All rays of a call form ONE reference chunk unless
    `chunk` is given (the view-direction tiling of model_pixel.py:219-222 depends on chunk membership)."""

    def __init__(self, num_levels=2, min_deg_point=0, max_deg_point=10, deg_view=4, num_coarse_samples=64,
                 num_fine_samples=64, use_viewdirs=True, noise_std=0.0, lindisp=False, num_src_views=3, encoder=None):
        super().__init__()
        if num_levels != 2 or not use_viewdirs or lindisp:
            raise NotImplementedError("only the reference's default 2-level, view-dependent, linear-depth setup")
        self.num_levels, self.num_src_views = num_levels, num_src_views
        self.num_coarse_samples, self.num_fine_samples = num_coarse_samples, num_fine_samples
        self.noise_std, self.lindisp = noise_std, lindisp
        if encoder is not None:
            self.encoder = encoder
        self.coarse_mlp = PixelNeRFMLP(min_deg_point, max_deg_point, deg_view)
        self.fine_mlp = PixelNeRFMLP(min_deg_point, max_deg_point, deg_view)
        self._scene_key = None
        self._scene_ctx = None
        # gather the latent pre-projected through each MLP's first layer (128 instead of 512 channels per tap, 41 % fewer
        # MACs per point-view; csrc/mlp_pix_h.hip).  False: the reference's operation order.
        self.preproject = os.environ.get("NEO360_PIX_PREPROJECT", "1") != "0"

    def _context(self, device):
        ctx = super()._context(device)
        if getattr(ctx, "_pix_preproject", None) != bool(self.preproject):
            _lib.check(ctx.lib.neo_pix_set_preproject(ctx.handle, int(bool(self.preproject))))
            ctx._pix_preproject = bool(self.preproject)
        return ctx

    def _sync_weights(self, ctx):
        for slot, mlp in enumerate((self.coarse_mlp, self.fine_mlp)):
            layers = mlp.ordered_layers()
            ws = [f32(l.weight.detach(), "weight") for l in layers]
            bs = [f32(l.bias.detach(), "bias") for l in layers]
            fp = _fingerprint(ws + bs)
            if ctx.uploaded.get(("pix", slot)) == fp:
                continue
            _lib.check(ctx.lib.neo_pix_upload_mlp(ctx.handle, slot, _ptr_table(ws), _ptr_table(bs), ctx.stream()))
            ctx.uploaded[("pix", slot)] = fp

    @torch.no_grad()
    def set_scene(self, latent, image_wh):
        """latent (NV,512,Hf,Wf) as the reference's SpatialEncoder leaves it in `.latent`; image_wh = (W,H) of
        the source images (model_pixel.py:176-177).  Re-laid out channels-last on the device, once.  The module keeps a weak
        reference to the tensor: a differentiable forward sends the latent's gradient there."""
        self._latent_src = weakref.ref(latent)
        latent = f32(latent, "latent")
        ctx = self._context(latent.device)
        NV, Cl, Hf, Wf = latent.shape
        _lib.check(ctx.lib.neo_pix_set_scene(ctx.handle, ptr(latent), NV, Cl, Hf, Wf, float(image_wh[0]),
                                             float(image_wh[1]), ctx.stream()))
        torch.cuda.current_stream(latent.device).synchronize()   # the caller may free its tensor
        self._scene_ctx = ctx
        self._scene_serial = getattr(self, "_scene_serial", 0) + 1

    def _ensure_scene(self, rays):
        enc = getattr(self, "encoder", None)
        if enc is None:
            if self._scene_ctx is None:
                raise _lib.NeoError("no scene latent: call set_scene(...) or attach an encoder module")
            return
        src = rays["src_imgs"]
        if self._scene_key is not None and self._scene_ctx is not None and self._scene_key.matches((src,), enc):
            return
        enc(src)
        self.set_scene(enc.latent, (src.shape[-1], src.shape[-2]))
        self._scene_key = _SceneKey((src,), enc)
        self.encoder_runs = getattr(self, "encoder_runs", 0) + 1

    _camera_args = NeRF_TP._camera_args

    @torch.no_grad()
    def eval_mlp(self, slot, rays, tvals, chunk=None):
        """Stage-level access for parity tests: latent lookups + pos_enc + MLP + activations at sample
        positions tvals (B,N) along rays_d.  Returns (B,N,4) = (sigmoid rgb, relu sigma)."""
        rays_o, rays_d, viewdirs = f32(rays["rays_o"]), f32(rays["rays_d"]), f32(rays["viewdirs"])
        tvals = f32(tvals, "tvals")
        ctx = self._context(rays_o.device)
        self._ensure_scene(rays)
        self._sync_weights(ctx)
        B, N = tvals.shape
        host_poses, NV, focal, cx, cy = self._camera_args(rays)
        out = torch.empty(B, N, 4, device=rays_o.device)
        _lib.check(ctx.lib.neo_pix_mlp(ctx.handle, slot, ptr(rays_o), ptr(rays_d), ptr(viewdirs), ptr(tvals), B, N,
                                       int(chunk or max(B, 1)), host_poses, NV, focal, cx, cy, ptr(out), ctx.stream()))
        self._raise_flags(ctx.poll_flags())     # stage-level access: always immediate
        return out

    def _scene_tensors_for_grad(self):
        ref = getattr(self, "_latent_src", None)
        return (ref(),) if ref is not None else ()

    def _latent_for_grad(self, rays):
        """The latent a differentiable call gathers from: an attached encoder is run WITH autograd on this batch, as the
        reference's forward does (model_pixel.py:176); otherwise the tensor given to set_scene, if the caller still holds it."""
        enc = getattr(self, "encoder", None)
        if enc is not None:
            src = rays["src_imgs"]
            enc(src)
            latent = enc.latent
            with torch.no_grad():
                self.set_scene(latent.detach(), (src.shape[-1], src.shape[-2]))
            self._scene_key = None
            return latent
        if self._scene_ctx is None:
            raise _lib.NeoError("no scene latent: call set_scene(...) or attach an encoder module")
        ref = getattr(self, "_latent_src", None)
        latent = ref() if ref is not None else None
        if latent is None:
            raise _lib.NeoError("a differentiable / randomized forward needs the latent tensor: attach an encoder module, or keep "
                                "the tensor passed to set_scene(...) alive (the module holds only a weak reference to it)")
        return latent

    def forward(self, rays, randomized, white_bkgd, near, far, chunk=None, seed=None):
        """randomized=True (stratified sampling, `noise_std`) or a call that wants gradients (`_wants_grad()`: the reference's
        training_step) runs on the operator chain of training.pix_render_train - every matrix product on the library's GEMMs,
        gradients to all 18 parameter tensors of each MLP and to the latent; otherwise the fused no-grad kernels."""
        if randomized or self._wants_grad(randomized):
            from . import training
            # A deterministic call that reached the chain by the AUTOMATIC rule alone keeps working where only the fused path can
            # serve it (a chunked frame; a set_scene latent the caller no longer holds): these calls worked before round 5 made
            # the chain the automatic choice - they fall back to the fused kernels instead of raising.
            fallback = not randomized and self._auto_chain()
            chunked = chunk is not None and chunk < rays["rays_o"].shape[0]
            if chunked and not fallback:
                raise NotImplementedError("the training call renders its rays as ONE reference chunk")
            latent = None
            if not chunked:
                try:
                    latent = self._latent_for_grad(rays)
                except _lib.NeoError:
                    if not fallback:
                        raise
            if latent is not None:
                return training.pix_render_train(self, rays, randomized, white_bkgd, near, far, latent, seed)
        with torch.no_grad():
            return self._forward_fused(rays, white_bkgd, near, far, chunk)

    def _forward_fused(self, rays, white_bkgd, near, far, chunk=None):
        raw = (rays["rays_o"], rays["rays_d"], rays["viewdirs"])
        rays_o = f32(raw[0], "rays_o")
        rays_d = f32(raw[1], "rays_d")
        viewdirs = f32(raw[2], "viewdirs")
        dev = rays_o.device
        ctx = self._context(dev)
        self._ensure_scene(rays)
        if self._scene_ctx is not ctx:
            raise _lib.NeoError("the scene latent was uploaded on a different device")
        self._sync_weights(ctx)
        self._before_call(ctx)
        B = rays_o.shape[0]
        host_poses, NV, focal, cx, cy = self._camera_args(rays)
        lv = []

        def launch():
            lv.extend((torch.empty(B, 3, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev)) for _ in range(2))
            _lib.check(ctx.lib.neo_pix_render(
                ctx.handle, ptr(rays_o), ptr(rays_d), ptr(viewdirs), B, int(chunk or max(B, 1)), host_poses, NV, focal, cx, cy,
                float(near), float(far), self.num_coarse_samples, self.num_fine_samples, int(bool(white_bkgd)),
                ptr(lv[0][0]), ptr(lv[0][1]), ptr(lv[0][2]), ptr(lv[1][0]), ptr(lv[1][1]), ptr(lv[1][2]), ctx.stream()))
            self._after_call(ctx)
            return [t for l in lv for t in l]
        self._launch_overlapped(ctx, dev, raw, (rays_o, rays_d, viewdirs), launch)
        return lv


class MipNeRF360MLP(nn.Module):
    """Parameter container with the layout of mipnerf360/model.py:30-107: pts_linear.0..depth-1
    (input = 504-d integrated positional encoding over the 21-direction geodesic basis, skip concat
    feeds index 5), density_layer and — unless disable_rgb — bottleneck_layer (256),
    views_linear.0 (128), rgb_layer; buffer pos_basis_t (3,21).  kaiming_uniform_ initialisers."""

    def __init__(self, netdepth=8, netwidth=256, disable_rgb=False):
        super().__init__()
        from .geopoly import icosahedron_basis
        self.netdepth, self.netwidth, self.disable_rgb = netdepth, netwidth, disable_rgb
        self.register_buffer("pos_basis_t", icosahedron_basis())
        pos = 12 * 2 * self.pos_basis_t.shape[-1]

        def lin(n_in, n_out):
            layer = nn.Linear(n_in, n_out)
            nn.init.kaiming_uniform_(layer.weight)
            return layer

        layers = [lin(pos, netwidth)]
        for idx in range(netdepth - 1):
            layers.append(lin(netwidth + pos if (idx % 4 == 0 and idx > 0) else netwidth, netwidth))
        self.pts_linear = nn.ModuleList(layers)
        self.density_layer = lin(netwidth, 1)
        if not disable_rgb:
            self.bottleneck_layer = lin(netwidth, 256)
            self.views_linear = nn.ModuleList([lin(256 + 27, 128)])
            self.rgb_layer = lin(128, 3)

    def ordered_layers(self):
        """Upload order fixed by include/neo360_hip.h (neo_mip_upload_mlp)."""
        out = list(self.pts_linear) + [self.density_layer]
        if not self.disable_rgb:
            out += [self.bottleneck_layer, self.views_linear[0], self.rgb_layer]
        return out


class MipNeRF360(_HipModule):
    """Mip-NeRF 360 renderer (mipnerf360/model.py:199-365): two proposal MLPs (4x256) + one NeRF MLP
    (8x1024).  `forward(batch, train_frac, randomized, is_train, near, far)` returns
    (renderings, ray_history) exactly as the reference: renderings[l] = {"rgb": (B,3)},
    ray_history[l] = dict(density (B,n), rgb (B,n,3), sdist (B,n+1), weights (B,n))."""

    def __init__(self, num_prop_samples=64, num_nerf_samples=32, num_levels=3, bg_intensity_range=(1.0, 1.0),
                 anneal_slope=10, stop_level_grad=True, use_viewdirs=True, ray_shape="cone",
                 disable_integration=False, single_jitter=True, dilation_multiplier=0.5, dilation_bias=0.0025,
                 num_glo_features=0, num_glo_embeddings=1000, learned_exposure_scaling=False, near_anneal_rate=None,
                 near_anneal_init=0.95, single_mlp=False, resample_padding=0.0, use_gpu_resampling=False,
                 opaque_background=True):
        super().__init__()
        if (num_levels, tuple(bg_intensity_range), anneal_slope, ray_shape, disable_integration, dilation_multiplier,
                dilation_bias, near_anneal_rate, resample_padding, opaque_background, use_viewdirs) != (
                3, (1.0, 1.0), 10, "cone", False, 0.5, 0.0025, None, 0.0, True, True):
            raise NotImplementedError("the HIP path implements the reference's default MipNeRF360 configuration")
        self.num_prop_samples, self.num_nerf_samples, self.num_levels = num_prop_samples, num_nerf_samples, num_levels
        self.mlps = nn.ModuleList([MipNeRF360MLP(4, 256, disable_rgb=True), MipNeRF360MLP(4, 256, disable_rgb=True),
                                   MipNeRF360MLP(8, 1024)])
        # NeRF MLP schedule (neo_mip_set_layered): None = the library's choice (layer-by-layer GEMMs from 8192 intervals per
        # call), True / False = always / never; $NEO360_MIP_LAYERED = 0 / 1 presets it (A/B runs of bench.py)
        self.layered = {"0": False, "1": True}.get(os.environ.get("NEO360_MIP_LAYERED", ""), None)

    def _context(self, device):
        ctx = super()._context(device)
        mode = -1 if self.layered is None else int(bool(self.layered))
        if getattr(ctx, "_mip_layered", None) != mode:
            _lib.check(ctx.lib.neo_mip_set_layered(ctx.handle, mode))
            ctx._mip_layered = mode
        return ctx

    def _sync_weights(self, ctx):
        for slot, mlp in enumerate(self.mlps):
            layers = mlp.ordered_layers()
            ws = [f32(l.weight.detach(), "weight") for l in layers]
            bs = [f32(l.bias.detach(), "bias") for l in layers]
            basis = f32(mlp.pos_basis_t, "pos_basis_t")
            fp = _fingerprint(ws + bs + [basis])
            if ctx.uploaded.get(("mip", slot)) == fp:
                continue
            _lib.check(ctx.lib.neo_mip_upload_mlp(ctx.handle, slot, mlp.netwidth, mlp.netdepth,
                                                  0 if mlp.disable_rgb else 1, _ptr_table(ws), _ptr_table(bs),
                                                  ptr(basis), ctx.stream()))
            ctx.uploaded[("mip", slot)] = fp

    def forward(self, batch, train_frac, randomized, is_train, near, far, seed=None):
        """randomized=True (one sampling jitter per ray and level) or a call that wants gradients (`_wants_grad()`: the
        reference's training_step, model.py:436-470) runs on the operator chain of training.mip_render_train - resampling,
        encodings and compositing native, every matrix product on the library's GEMMs, gradients to all parameters through the
        colours and the interval weights; otherwise the fused no-grad kernels.  `is_train` only selects how the reference
        evaluates the contraction's Jacobian (helper.py:45-66); both of its forms are the closed form used here."""
        if randomized or self._wants_grad(randomized):
            from . import training
            return training.mip_render_train(self, batch, train_frac, randomized, near, far, seed)
        with torch.no_grad():
            return self._forward_fused(batch, train_frac, near, far)

    def _forward_fused(self, batch, train_frac, near, far):
        raw = (batch["rays_o"], batch["rays_d"], batch["viewdirs"], batch["radii"])
        rays_o, rays_d = f32(raw[0], "rays_o"), f32(raw[1], "rays_d")
        viewdirs, radii = f32(raw[2], "viewdirs"), f32(raw[3], "radii")
        dev = rays_o.device
        ctx = self._context(dev)
        self._sync_weights(ctx)
        self._before_call(ctx)
        B = rays_o.shape[0]
        counts = (self.num_prop_samples, self.num_prop_samples, self.num_nerf_samples)
        bufs = []

        def launch():
            bufs.extend(dict(rgb=torch.empty(B, 3, device=dev), sdist=torch.empty(B, n + 1, device=dev),
                             weights=torch.empty(B, n, device=dev), rgbdens=torch.empty(B, n, 4, device=dev)) for n in counts)
            arr = (_lib.MipLevelOut * 3)(*[_lib.MipLevelOut(*(b[k].data_ptr() for k in ("rgb", "sdist", "weights", "rgbdens")))
                                           for b in bufs])
            _lib.check(ctx.lib.neo_mip_render(ctx.handle, ptr(rays_o), ptr(rays_d), ptr(viewdirs), ptr(radii), B,
                                              float(train_frac), float(near), float(far), self.num_prop_samples,
                                              self.num_nerf_samples, arr, ctx.stream()))
            self._after_call(ctx)
            return [t for b in bufs for t in b.values()]
        self._launch_overlapped(ctx, dev, raw, (rays_o, rays_d, viewdirs, radii), launch)
        renderings = [{"rgb": b["rgb"]} for b in bufs]
        history = [dict(density=b["rgbdens"][..., 3], rgb=b["rgbdens"][..., :3], sdist=b["sdist"], weights=b["weights"])
                   for b in bufs]
        return renderings, history

    @torch.no_grad()
    def eval_mlp(self, slot, batch, tdist):
        """Stage-level access: cast_rays + MLP for the intervals of tdist (B,n+1) -> (B,n,4) = (rgb, density)."""
        rays_o, rays_d = f32(batch["rays_o"]), f32(batch["rays_d"])
        viewdirs, radii, tdist = f32(batch["viewdirs"]), f32(batch["radii"]), f32(tdist)
        ctx = self._context(rays_o.device)
        self._sync_weights(ctx)
        B, n1 = tdist.shape
        out = torch.empty(B, n1 - 1, 4, device=rays_o.device)
        _lib.check(ctx.lib.neo_mip_mlp(ctx.handle, slot, ptr(rays_o), ptr(rays_d), ptr(viewdirs), ptr(radii), ptr(tdist),
                                       B, n1 - 1, ptr(out), ctx.stream()))
        self._raise_flags(ctx.poll_flags())     # stage-level access: always immediate
        return out
