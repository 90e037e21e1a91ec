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

import torch
import torch.nn as nn

from . import _lib
from .context import f32, new_context, ptr


def _xavier_linear(n_in, n_out, xavier=True):
    layer = nn.Linear(n_in, n_out)
    if xavier:
        nn.init.xavier_uniform_(layer.weight)
    return layer


def _fingerprint(tensors):
    return tuple((t.data_ptr(), t._version, t.device) for t in tensors)


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

    def _context(self, device):
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
        ctx = self._ctx_cache.get(key)
        if ctx is None:
            ctx = self._ctx_cache[key] = new_context(device)
        return ctx

    @staticmethod
    def _check_mode(randomized):
        if randomized:
            raise NotImplementedError(
                "randomized=True (stratified jitter / training) is outside the accelerated inference path")


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

    @torch.no_grad()
    def forward(self, rays, randomized, white_bkgd, near, far):
        """Returns [(rgb (B,3), acc (B,), depth (B,))] * 2, as the reference."""
        self._check_mode(randomized)
        rays_o = f32(rays["rays_o"], "rays_o")
        viewdirs = f32(rays["viewdirs"], "viewdirs")
        rays_d = f32(rays["rays_d"], "rays_d")
        dev = rays_o.device
        ctx = self._context(dev)
        self._sync_weights(ctx)
        B = rays_o.shape[0]
        outs = [(torch.empty(B, 3, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev))
                for _ in range(2)]
        _lib.check(ctx.lib.neo_vanilla_render(
            ctx.handle, ptr(rays_o), ptr(viewdirs), ptr(rays_d), B, float(near), float(far),
            self.num_coarse_samples, self.num_fine_samples, int(bool(white_bkgd)),
            ptr(outs[0][0]), ptr(outs[0][1]), ptr(outs[0][2]), ptr(outs[1][0]), ptr(outs[1][1]), ptr(outs[1][2]),
            ctx.stream()))
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
        return out
