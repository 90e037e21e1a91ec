"""Training-side operators as PyTorch autograd functions over the C ABI:

  composite(...)        volumetric_rendering with a HIP backward (neo360/helper.py:128-171, vanilla :521-559)
  gather_features(...)  index_grid + get_local_feats as one op, backward = scatter-add into the feature maps
  eff_distloss(w, m, interval)   torch_efficient_distloss.eff_distloss (call site neo360/model.py:1246-1260)
  rand_uniform / sample_level0 / resample_u   the randomized=True samplers on a counter-based generator

The fused MLP evaluators are inference kernels: these ops cover the parts of the training step around them
(sampling, lookups, compositing, regulariser); `models.NeRF_TP(...)(rays, randomized, white_bkgd, near, far)` returns
the reference's training tuple (forward values) through `neo_tp_render_train`.
"""
import ctypes
import math
import os

import torch

from . import _lib
from .context import f32, get_context, ptr


def _ctx(t, ctx):
    return ctx if ctx is not None else get_context(t.device)


def rand_uniform(seed, stream_id, rows, cols, device="cuda", ctx=None):
    """(rows, cols) uniforms in [0,1): Philox4x32-10 keyed by `seed`, counter (row, col, stream_id, 0), 24 bits."""
    ctx = ctx if ctx is not None else get_context(device)
    out = torch.empty(rows, cols, device=ctx.device)
    _lib.check(ctx.lib.neo_rand_uniform(ctx.handle, int(seed), int(stream_id), rows, cols, ptr(out), ctx.stream()))
    return out


def sample_level0(far, n_coarse, u_fg=None, u_bg=None, ctx=None):
    """neo360/helper.py:24-75 for both regions.  far (R,1) or (R,); u_* (R, n_coarse+1) uniforms = randomized=True.
    Returns fg_t (ascending t) and bg_s (descending inverse radius), (R, n_coarse+1) each."""
    far = f32(far, "far").reshape(-1)
    ctx = _ctx(far, ctx)
    R = far.shape[0]
    fg = torch.empty(R, n_coarse + 1, device=far.device)
    bg = torch.empty(R, n_coarse + 1, device=far.device)
    u_fg = f32(u_fg, "u_fg") if u_fg is not None else None
    u_bg = f32(u_bg, "u_bg") if u_bg is not None else None
    _lib.check(ctx.lib.neo_tp_sample_level0(ctx.handle, ptr(far), R, n_coarse, ptr(u_fg), ptr(u_bg), ptr(fg), ptr(bg),
                                            ctx.stream()))
    return fg, bg


def resample_u(t_prev, weights, u, descending=False, ctx=None):
    """sorted_piecewise_constant_pdf(randomized=True) + the sort of sample_pdf with the caller's uniforms u (R, n_new)."""
    t_prev, weights, u = f32(t_prev, "t_prev"), f32(weights, "weights").detach(), f32(u, "u")
    ctx = _ctx(t_prev, ctx)
    R, n_prev = t_prev.shape
    n_new = u.shape[1]
    out = torch.empty(R, n_prev + n_new, device=t_prev.device)
    _lib.check(ctx.lib.neo_resample_u(ctx.handle, ptr(t_prev), ptr(weights), ptr(u), R, n_prev, n_new, int(descending),
                                      ptr(out), ctx.stream()))
    return out


class _Composite(torch.autograd.Function):
    @staticmethod
    def forward(ctx_, mode, rgb, sigma, t, rays_d, t_far, white_bkgd, lib_ctx):
        packed = sigma is None                  # rgb is the packed (R,N,4) = (rgb, sigma) tensor of `activate`
        if packed:
            rgbsigma = f32(rgb, "rgbsigma")
        else:
            rgbsigma = torch.cat([f32(rgb, "rgb"), f32(sigma, "sigma").reshape(*rgb.shape[:-1], 1)], dim=-1).contiguous()
        t = f32(t, "t")
        c = _ctx(t, lib_ctx)
        R, N = t.shape
        dev = t.device
        rays_d = f32(rays_d, "rays_d") if rays_d is not None else None
        t_far = f32(t_far, "t_far").reshape(-1) if t_far is not None else None
        out_rgb, acc, depth = torch.empty(R, 3, device=dev), torch.empty(R, device=dev), torch.empty(R, device=dev)
        w = torch.empty(R, N, device=dev)
        lam = torch.empty(R, 1, device=dev) if mode == 1 else torch.zeros(R, 1, device=dev)
        _lib.check(c.lib.neo_composite(c.handle, mode, ptr(rgbsigma), ptr(t), ptr(rays_d), ptr(t_far), R, N,
                                       int(bool(white_bkgd)), ptr(out_rgb), ptr(acc), ptr(depth), ptr(w),
                                       ptr(lam) if mode == 1 else None, c.stream()))
        ctx_.save_for_backward(rgbsigma, t, rays_d if rays_d is not None else torch.empty(0, device=dev),
                               t_far if t_far is not None else torch.empty(0, device=dev))
        ctx_.meta = (mode, bool(white_bkgd), c, rays_d is not None, t_far is not None, None if packed else tuple(sigma.shape))
        return out_rgb, acc, w, lam, depth

    @staticmethod
    def backward(ctx_, g_rgb, g_acc, g_w, g_lam, g_depth):
        rgbsigma, t, rays_d, t_far = ctx_.saved_tensors
        mode, white, c, has_d, has_far, sigma_shape = ctx_.meta
        need_rgb, need_sigma = ctx_.needs_input_grad[1], ctx_.needs_input_grad[2] or sigma_shape is None
        if not (need_rgb or need_sigma):
            return (None,) * 8
        R, N = t.shape
        g = torch.empty(R, N, 4, device=t.device)
        cont = lambda x: f32(x.contiguous(), "grad") if x is not None else None
        g_rgb, g_acc, g_w, g_lam, g_depth = cont(g_rgb), cont(g_acc), cont(g_w), cont(g_lam), cont(g_depth)
        _lib.check(c.lib.neo_composite_backward(c.handle, mode, ptr(rgbsigma), ptr(t), ptr(rays_d) if has_d else None,
                                                ptr(t_far) if has_far else None, R, N, int(white), ptr(g_rgb), ptr(g_acc),
                                                ptr(g_depth), ptr(g_w), ptr(g_lam), ptr(g), c.stream()))
        if sigma_shape is None:                 # packed input: one gradient tensor
            return (None, g, None, None, None, None, None, None)
        # sigma may have been (R,N) or (R,N,1) in the forward: its gradient takes that shape
        return (None, g[..., :3] if need_rgb else None, g[..., 3].reshape(sigma_shape) if need_sigma else None,
                None, None, None, None, None)


def composite(mode, rgb, sigma, t, rays_d=None, t_far=None, white_bkgd=False, ctx=None):
    """Differentiable volumetric_rendering.  mode 0 vanilla, 1 NeO-360 inside the sphere, 2 outside.
    rgb (R,N,3), sigma (R,N,1) -> (comp_rgb (R,3), acc (R,), weights (R,N), bg_lambda (R,1), depth (R,));
    gradients flow to rgb and sigma (sample positions are detached in the reference, helper.py:224).
    sigma=None: `rgb` is the packed (R,N,4) = (rgb, sigma) tensor `activate` returns (no concatenation)."""
    return _Composite.apply(mode, rgb, sigma, t, rays_d, t_far, white_bkgd, ctx)


class _Activate(torch.autograd.Function):
    @staticmethod
    def forward(ctx_, raw_rgb, raw_sigma, noise, noise_scale, lib_ctx):
        raw_rgb, raw_sigma = f32(raw_rgb, "raw_rgb"), f32(raw_sigma, "raw_sigma")
        P = raw_sigma.numel()
        if raw_rgb.numel() != 3 * P:
            raise ValueError("raw_rgb must hold 3 values per raw_sigma entry, got %s / %s" % (tuple(raw_rgb.shape), tuple(raw_sigma.shape)))
        noise = f32(noise, "noise") if noise is not None else None
        if noise is not None and noise.numel() != P:
            raise ValueError("noise must hold one value per point")
        c = _ctx(raw_sigma, lib_ctx)
        out = torch.empty(P, 4, device=raw_sigma.device)
        _lib.check(c.lib.neo_tp_activate(c.handle, ptr(raw_rgb), ptr(raw_sigma), ptr(noise), float(noise_scale), P, ptr(out), c.stream()))
        ctx_.save_for_backward(raw_rgb, raw_sigma, noise if noise is not None else torch.empty(0, device=out.device))
        ctx_.meta = (c, float(noise_scale), noise is not None, tuple(raw_rgb.shape), tuple(raw_sigma.shape))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx_, g):
        raw_rgb, raw_sigma, noise = ctx_.saved_tensors
        c, scale, has_noise, shape_rgb, shape_sigma = ctx_.meta
        P = raw_sigma.numel()
        g = f32(g.reshape(P, 4).contiguous(), "grad")
        g_rgb, g_sigma = torch.empty(P, 3, device=g.device), torch.empty(P, device=g.device)
        _lib.check(c.lib.neo_tp_activate_backward(c.handle, ptr(raw_rgb), ptr(raw_sigma), ptr(noise) if has_noise else None, scale, P,
                                                  ptr(g), ptr(g_rgb), ptr(g_sigma), c.stream()))
        return g_rgb.reshape(shape_rgb), g_sigma.reshape(shape_sigma), None, None, None


def activate(raw_rgb, raw_sigma, noise=None, noise_scale=0.0, ctx=None):
    """The reference's output activations (neo360/model.py:380-385, vanilla_nerf/model.py:194-205) as ONE op with a native backward:
    (P,4) = (sigmoid(raw_rgb) * 1.002 - 0.001, softplus(raw_sigma + noise * noise_scale - 1)); noise (P values) = the uniforms
    of `density_noise` / `noise_std`.  The packed result feeds `composite(mode, packed, None, ...)`."""
    return _Activate.apply(raw_rgb, raw_sigma, noise, noise_scale, ctx)


def train_points(module, region, rays_o, rays_d, tvals, far, poses, ctx=None):
    """Sample points of one region (0 inside / 1 outside the sphere) for the training call: look (R*N,3) world points for the
    feature lookups and x_enc (NV, R*N, 63 | 84) reference-order encodings of the camera-frame point per source view
    (neo_tp_train_points: the evaluators' own per-point set-up - neo360/helper.py:24-75, :401-451, util.py:52-70 - so the
    operator chain and the fused kernels see bitwise the same points).  No gradients: sample positions are detached in the
    reference (helper.py:224) and the rays / poses are data."""
    rays_o, rays_d, tvals = f32(rays_o, "rays_o"), f32(rays_d, "rays_d"), f32(tvals, "tvals")
    c = _ctx(rays_o, ctx)
    R, N = tvals.shape
    poses = f32(poses, "src_poses")
    NV = poses.shape[0]
    host = poses.detach().cpu().contiguous()
    host_poses = (ctypes.c_float * (16 * NV))(*host.reshape(-1).tolist())
    ch = 4 if region else 3
    look = torch.empty(R * N, 3, device=rays_o.device)
    x_enc = torch.empty(NV, R * N, 21 * ch, device=rays_o.device)
    far = f32(far, "far").reshape(-1) if far is not None else None
    _lib.check(c.lib.neo_tp_train_points(c.handle, ch, ptr(rays_o), ptr(rays_d), ptr(tvals), ptr(far), R, N, host_poses, NV,
                                         ptr(look), ptr(x_enc), c.stream()))
    return look, x_enc


class _DistLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx_, w, m, interval, lib_ctx):
        w2, m2 = f32(w, "w").reshape(-1, w.shape[-1]), f32(m, "m").reshape(-1, w.shape[-1])
        c = _ctx(w2, lib_ctx)
        R, N = w2.shape
        loss = torch.empty(R, device=w2.device)
        grad = torch.empty(R, N, device=w2.device)
        _lib.check(c.lib.neo_distloss(c.handle, ptr(w2), ptr(m2), R, N, float(interval), ptr(loss), ptr(grad), c.stream()))
        ctx_.save_for_backward(grad)
        ctx_.shape = (w.shape, R)
        return loss.sum() / R

    @staticmethod
    def backward(ctx_, g):
        (grad,) = ctx_.saved_tensors
        shape, R = ctx_.shape
        return (grad * (g / R)).reshape(shape), None, None, None


def eff_distloss(w, m, interval, ctx=None):
    """torch_efficient_distloss.eff_distloss: w (..., N) weights, m (..., N) interval midpoints, scalar interval."""
    return _DistLoss.apply(w, m, interval, ctx)


class _Gather(torch.autograd.Function):
    @staticmethod
    def forward(ctx_, module, pts, plane_xz, plane_xy, plane_yz, latent, rays, planes_only=False, shared=None):
        # the module's context holds channels-last copies of the scene: they must be copies of THESE tensors at THIS
        # version (an optimizer step or a fresh encoder output would otherwise leave the forward values stale while
        # gradients still flow to the arguments) - re-upload when the fingerprint differs
        pts = f32(pts, "pts").reshape(-1, 3)
        maps = (plane_xz, plane_xy, plane_yz, latent)
        if not module.scene_matches(maps):
            module.set_scene(plane_xz.detach(), plane_xy.detach(), plane_yz.detach(), latent.detach(),
                             module.scene_image_wh(rays), _source=maps)
        c = module._context(pts.device)
        host_poses, NV, focal, cx, cy = module._camera_args(rays)
        P = pts.shape[0]
        world = torch.empty(NV * P, 128, device=pts.device)
        local = None if planes_only else torch.empty(NV * P, 512, device=pts.device)
        _lib.check(c.lib.neo_tp_gather(c.handle, ptr(pts), P, host_poses, NV, focal, cx, cy, ptr(world), ptr(local), c.stream()))
        ctx_.save_for_backward(pts)
        ctx_.meta = (c, host_poses, NV, focal, cx, cy, plane_xz.shape, latent.shape, bool(planes_only), shared)
        if shared is not None:
            shared["uses"] = shared.get("uses", 0) + 1
        if planes_only:
            return world, torch.empty(0, device=pts.device)
        return world, local

    @staticmethod
    def backward(ctx_, g_world, g_local):
        (pts,) = ctx_.saved_tensors
        c, host_poses, NV, focal, cx, cy, pshape, lshape, planes_only, shared = ctx_.meta
        dev = pts.device
        _, Cw, Hp, Wp = pshape
        _, Cl, Hf, Wf = lshape
        # `shared` (round 6): the lookups of one training call scatter into ONE set of channels-last plane-gradient buffers - the first
        # backward to run allocates them and hands them to autograd, the others add into them and return nothing: one zero fill, one
        # NHWC -> NCHW conversion and no full-size sums per plane and call instead of one of each per lookup (4 per call)
        first = True
        if shared is not None and planes_only:
            first = shared.get("planes") is None
            if first:
                shared["planes"] = [torch.zeros(NV, Hp, Wp, Cw, device=dev) for _ in range(3)]
                shared["left"] = shared.get("uses", 1)
            gp = shared["planes"]
            shared["left"] -= 1
            if shared["left"] <= 0:                                      # every lookup of the call has reported: a later pass starts afresh
                shared["planes"] = None
        else:
            gp = [torch.zeros(NV, Hp, Wp, Cw, device=dev) for _ in range(3)]
        gl = None if planes_only else torch.zeros(NV, Hf, Wf, Cl, device=dev)
        gw = f32(g_world.contiguous(), "g_world")
        gloc = None if planes_only else f32(g_local.contiguous(), "g_local")
        _lib.check(c.lib.neo_tp_gather_backward(c.handle, ptr(pts), pts.shape[0], host_poses, NV, focal, cx, cy, ptr(gw),
                                                ptr(gloc), ptr(gp[0]), ptr(gp[1]), ptr(gp[2]), ptr(gl), c.stream()))
        nchw = lambda x: x.permute(0, 3, 1, 2)
        if not first:
            return None, None, None, None, None, None, None, None, None
        return None, None, nchw(gp[0]), nchw(gp[1]), nchw(gp[2]), (nchw(gl) if gl is not None else None), None, None, None


def gather_features(module, pts, plane_xz, plane_xy, plane_yz, latent, rays):
    """index_grid (three tri-planes summed) + get_local_feats at world points pts (P,3) for every source view of `rays`
    (src_poses / src_focal / src_c): world (NV*P,128), local (NV*P,512), view-major rows.  `module` is the NeRF_TP that
    holds the device-side copies: when they are not copies of these very tensors at their current version (fingerprint
    recorded by `module.set_scene`), the four maps are uploaded again first (image size from the previous `set_scene` or
    `rays["src_imgs"]`), so forward values and gradients always refer to the same data.  Gradients flow to the four
    feature maps (NCHW, like the inputs)."""
    return _Gather.apply(module, pts, plane_xz, plane_xy, plane_yz, latent, rays)


def gather_planes(module, pts, plane_xz, plane_xy, plane_yz, latent, rays, shared=None):
    """The tri-plane half of gather_features alone: world (NV*P,128).  The latent is not looked up (the projected-space
    training path gathers it through `gather_map`); it is still passed so that the device-side scene - geometry included -
    follows these tensors."""
    return _Gather.apply(module, pts, plane_xz, plane_xy, plane_yz, latent, rays, True, shared)[0]


class _ChannelsLast(torch.autograd.Function):
    """(NV, C, H, W) -> (NV H W, C) rows for the texel-space GEMMs, and the gradient back, each as one tiled transpose
    (neo_transpose) - torch's permute + reshape is a strided copy both ways (0.66 + 1.04 ms of a 31 ms step for the 472 MB latent)."""

    @staticmethod
    def forward(ctx_, x, lib_ctx):
        x = f32(x, "map")
        if x.dim() != 4:
            raise ValueError("expected an NCHW map, got %s" % (tuple(x.shape),))
        if not x.is_contiguous():
            x = x.contiguous()
        nv, ch, h, w = x.shape
        c = _ctx(x, lib_ctx)
        out = torch.empty(nv * h * w, ch, device=x.device)
        _lib.check(c.lib.neo_transpose(c.handle, ptr(x), nv, ch, h * w, ptr(out), c.stream()))
        ctx_.meta = (c, nv, ch, h, w)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx_, g):
        c, nv, ch, h, w = ctx_.meta
        g = f32(g.contiguous(), "g")
        out = torch.empty(nv, ch, h, w, device=g.device)
        _lib.check(c.lib.neo_transpose(c.handle, ptr(g), nv, h * w, ch, ptr(out), c.stream()))
        return out, None


def channels_last_rows(x, ctx=None):
    """An NCHW feature map as (NV H W, C) rows, differentiable (neo_transpose both ways)."""
    return _ChannelsLast.apply(x, ctx)


class _MapGather(torch.autograd.Function):
    """Bilinear lookup in a caller-owned channels-last map (NV Hf Wf, C) at get_local_feats' taps, with the run-merged
    scatter-add as its backward (neo_tp_gather_map / _backward).  `col` / `width`: a column slice of a wider map (round 6:
    neo_tp_gather_map_slice) whose gradient is scattered into the matching slice of ONE buffer shared by every lookup in that map
    (`shared`, see project_latent_all): the first lookup to run its backward hands the buffer to autograd, the others add into it."""

    @staticmethod
    def forward(ctx_, module, gmap, pts, rays, kind="tp", col=None, width=None, shared=None):
        pts = f32(pts, "pts").reshape(-1, 3)
        gmap = f32(gmap, "map")
        if gmap.dim() != 2 or gmap.shape[1] % 64 != 0:
            raise ValueError("map must be (texels, C) with C a multiple of 64, got %s" % (tuple(gmap.shape),))
        c = module._context(pts.device)
        host_poses, NV, focal, cx, cy = module._camera_args(rays)
        P = pts.shape[0]
        sliced = col is not None
        if sliced:
            if kind != "tp" or width % 64 or col % 4 or col < 0 or col + width > gmap.shape[1]:
                raise ValueError("bad map slice [%s, +%s) of %s (kind %s)" % (col, width, tuple(gmap.shape), kind))
            C = int(width)
        else:
            C = gmap.shape[1]
        out = torch.empty(NV * P, C, device=pts.device)
        # the row count travels with the pointer: the library rejects a map that is not NV*Hf*Wf rows of the geometry uploaded in
        # this module's context (it used to be indexed - and, in the backward, scattered into - unchecked)
        if sliced:
            _lib.check(c.lib.neo_tp_gather_map_slice(c.handle, gmap.data_ptr() + 4 * int(col), gmap.shape[0], gmap.shape[1], C, ptr(pts), P,
                                                     host_poses, NV, focal, cx, cy, ptr(out), c.stream()))
        else:
            fn = c.lib.neo_pix_gather_map if kind == "pix" else c.lib.neo_tp_gather_map
            _lib.check(fn(c.handle, ptr(gmap), gmap.shape[0], C, ptr(pts), P, host_poses, NV, focal, cx, cy, ptr(out), c.stream()))
        ctx_.save_for_backward(pts)
        ctx_.meta = (c, host_poses, NV, focal, cx, cy, tuple(gmap.shape), kind, col, C, shared)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx_, g_out):
        (pts,) = ctx_.saved_tensors
        c, host_poses, NV, focal, cx, cy, mshape, kind, col, C, shared = ctx_.meta
        g = f32(g_out.contiguous(), "g_out")
        if col is not None:
            # shared = None (module.train_shared_grads = False): a private full-size buffer per lookup, summed by autograd - the plain
            # contract of a custom Function, without the in-place accumulation into a tensor already handed to the engine
            first = shared is None or shared.get("buf") is None
            if shared is None:
                buf = torch.zeros(mshape, device=pts.device)
            else:
                if first:
                    shared["buf"] = torch.zeros(mshape, device=pts.device)
                buf = shared["buf"]
            _lib.check(c.lib.neo_tp_gather_map_slice_backward(c.handle, mshape[0], mshape[1], C, ptr(pts), pts.shape[0], host_poses, NV, focal,
                                                              cx, cy, ptr(g), buf.data_ptr() + 4 * int(col), c.stream()))
            return None, (buf if first else None), None, None, None, None, None, None
        g_map = torch.zeros(mshape, device=pts.device)
        fn = c.lib.neo_pix_gather_map_backward if kind == "pix" else c.lib.neo_tp_gather_map_backward
        _lib.check(fn(c.handle, mshape[0], mshape[1], ptr(pts), pts.shape[0], host_poses, NV, focal, cx, cy, ptr(g), ptr(g_map), c.stream()))
        return None, g_map, None, None, None, None, None, None


class _GradSink(torch.autograd.Function):
    """Identity between a merged projected map and its lookups: its backward runs once, after every lookup has scattered into the shared
    gradient buffer, and releases the buffer so that a second backward through the same graph starts from zeros again."""

    @staticmethod
    def forward(ctx_, gmap, shared):
        ctx_.shared = shared
        return gmap.view_as(gmap)

    @staticmethod
    def backward(ctx_, g):
        ctx_.shared["buf"] = None
        return g, None


def gather_map(module, gmap, pts, rays, kind="tp", col=None, width=None, shared=None):
    """Lookup of world points pts (P,3) in a channels-last map gmap (NV*Hf*Wf, C) with the latent's geometry, every source view:
    (NV*P, C) view-major rows.  kind "tp": NeRF_TP's get_local_feats taps (scene of neo_tp_set_scene); "pix": the PixelNeRF decoder's
    (neo_pix_set_scene).  Differentiable w.r.t. the map.  col / width / shared: columns [col, col + width) of a merged map made by
    project_latent_all (which also returns `shared`)."""
    return _MapGather.apply(module, gmap, pts, rays, kind, col, width, shared)


def _zeros_like_shapes(wshapes, bshapes, dev):
    """Zeroed weight / bias gradient tensors of a chain as views of ONE buffer: one fill kernel instead of 18 (each view starts at a
    multiple of 4 floats)."""
    shapes = list(wshapes) + list(bshapes)
    sizes = [int(math.prod(sh)) for sh in shapes]
    offs, total = [], 0
    for n in sizes:
        offs.append(total)
        total += (n + 3) // 4 * 4
    flat = torch.zeros(total, device=dev)
    views = [flat[o:o + n].view(sh) for o, n, sh in zip(offs, sizes, shapes)]
    return views[:len(wshapes)], views[len(wshapes):]


class _TrainMLPPre(torch.autograd.Function):
    """NeRFPPMLP on the PROJECTED latent: `pre` (NV*P, 256) = [W0_loc f | W3_loc f] replaces the (NV*P, 512) local feature rows
    (neo_tp_mlp_train_forward_pre / _backward_pre).  The gradient of the local weight columns is NOT produced here: it flows
    through `pre` into the texel-space GEMM that made the projected map."""

    @staticmethod
    def forward(ctx_, lib_ctx, input_ch, nv, x_enc, cond_rows, world_feat, pre, *params):
        ws, bs = params[:9], params[9:]
        pe, npts = input_ch * 21, x_enc.shape[1]
        if x_enc.dim() != 3 or tuple(x_enc.shape) != (nv, npts, pe):
            raise ValueError("x_enc must be (NV, P, %d), got %s" % (pe, tuple(x_enc.shape)))
        for name, t, width in (("cond_rows", cond_rows, 27), ("world_feat", world_feat, 128), ("pre", pre, 256)):
            if tuple(t.shape) != (nv * npts, width):
                raise ValueError("%s must be (NV*P, %d) = (%d, %d), got %s" % (name, width, nv * npts, width, tuple(t.shape)))
        want = [(128, pe + 640), (128, 128), (128, 128), (128, 128 + pe + 640), (64, 155), (64, 64), (128, 128), (1, 128), (3, 64)]
        for i, (w, b) in enumerate(zip(ws, bs)):
            if tuple(w.shape) != want[i] or tuple(b.shape) != (want[i][0],):
                raise ValueError("layer %d: weight %s / bias %s, expected %s / (%d,)"
                                 % (i, tuple(w.shape), tuple(b.shape), want[i], want[i][0]))
        xe = f32(x_enc, "x_enc").reshape(-1, pe)
        pf, wf, cond = f32(pre, "pre"), f32(world_feat, "world_feat"), f32(cond_rows, "cond_rows")
        c = _ctx(xe, lib_ctx)
        wd = [f32(w.detach(), "weight") for w in ws]
        bd = [f32(b.detach(), "bias") for b in bs]
        tab = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        raw_rgb = torch.empty(npts, 3, device=xe.device)
        raw_sigma = torch.empty(npts, 1, device=xe.device)
        tape = torch.empty(c.lib.neo_tp_mlp_train_tape_floats(nv, npts), device=xe.device)
        _lib.check(c.lib.neo_tp_mlp_train_forward_pre(c.handle, input_ch, tab(wd), tab(bd), ptr(xe), ptr(pf), ptr(wf), ptr(cond), nv,
                                                      npts, ptr(tape), ptr(raw_rgb), ptr(raw_sigma), c.stream()))
        ctx_.save_for_backward(xe, wf, cond, tape, *wd)
        ctx_.meta = (c, input_ch, nv, npts, x_enc.shape, [tuple(w.shape) for w in ws], [tuple(b.shape) for b in bs])
        return raw_rgb, raw_sigma

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx_, g_rgb, g_sigma):
        xe, wf, cond, tape, *wd = ctx_.saved_tensors
        c, input_ch, nv, npts, xshape, wshapes, bshapes = ctx_.meta
        if ctx_.needs_input_grad[4]:
            raise NotImplementedError("nerfpp_mlp: no gradient for cond_rows (the reference's view directions are data)")
        dev = xe.device
        g_rgb = f32(g_rgb.contiguous(), "g_rgb") if g_rgb is not None else torch.zeros(npts, 3, device=dev)
        g_sigma = f32(g_sigma.contiguous(), "g_sigma") if g_sigma is not None else torch.zeros(npts, 1, device=dev)
        gw, gb = _zeros_like_shapes(wshapes, bshapes, dev)
        gx = torch.empty_like(xe) if ctx_.needs_input_grad[3] else None
        gworld = torch.empty_like(wf) if ctx_.needs_input_grad[5] else None
        gpre = torch.empty(nv * npts, 256, device=dev)
        tab = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        _lib.check(c.lib.neo_tp_mlp_train_backward_pre(c.handle, input_ch, tab(wd), ptr(xe), ptr(wf), ptr(cond), nv, npts, ptr(tape),
                                                       ptr(g_rgb), ptr(g_sigma), tab(gw), tab(gb), ptr(gx), ptr(gpre), ptr(gworld),
                                                       c.stream()))
        return (None, None, None, gx.reshape(xshape) if gx is not None else None, None, gworld, gpre, *gw, *gb)


class _Linear(torch.autograd.Function):
    """y = x W^T + b [ReLU] on the library's exact-fp32 GEMM (neo_linear_forward); backward = neo_linear_input_grad and
    neo_linear_weight_grad.  W may be a column block of a wider matrix (its row pitch is passed on)."""

    @staticmethod
    def forward(ctx_, lib_ctx, x, w, b, relu):
        x = f32(x, "x")
        if x.dim() != 2 or w.dim() != 2 or x.shape[1] != w.shape[1]:
            raise ValueError("x (rows, in) and W (out, in) do not match: %s, %s" % (tuple(x.shape), tuple(w.shape)))
        c = _ctx(x, lib_ctx)
        wd = w.detach()
        if wd.dtype != torch.float32 or wd.stride(1) != 1:
            wd = wd.float().contiguous()
        rows, out_f, in_f = x.shape[0], w.shape[0], w.shape[1]
        y = torch.empty(rows, out_f, device=x.device)
        bd = f32(b.detach(), "bias") if b is not None else None
        _lib.check(c.lib.neo_linear_forward(c.handle, rows, out_f, in_f, ptr(x), in_f, ptr(wd), wd.stride(0), ptr(bd) if bd is not None else None,
                                            int(bool(relu)), 0, ptr(y), out_f, c.stream()))
        ctx_.save_for_backward(x, wd, y if relu else None)
        ctx_.meta = (c, bool(relu), b is not None)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx_, gy):
        x, wd, y = ctx_.saved_tensors
        c, relu, has_b = ctx_.meta
        g = gy.contiguous()
        if relu:
            g = g * (y > 0)
        rows, out_f, in_f = x.shape[0], wd.shape[0], wd.shape[1]
        gx = gw = gb = None
        if ctx_.needs_input_grad[1]:
            gx = torch.empty(rows, in_f, device=x.device)
            _lib.check(c.lib.neo_linear_input_grad(c.handle, rows, in_f, out_f, ptr(g), out_f, ptr(wd), wd.stride(0), 0, ptr(gx), in_f, c.stream()))
        if ctx_.needs_input_grad[2] or (has_b and ctx_.needs_input_grad[3]):
            gw, gb = weight_grad(g, x, bias=True, ctx=c)
            if not has_b:
                gb = None
        return None, gx, gw, gb, None


def linear(x, w, b=None, relu=False, ctx=None):
    """torch.nn.functional.linear (+ ReLU) with the library's GEMMs forward and backward: x (rows, in), W (out, in) - possibly a
    column block `W[:, a:b]` of a wider weight -, b (out) or None."""
    return _Linear.apply(ctx, x, w, b, relu)


class _LinearNoBias:
    """y = x W^T for a texel-space projection (230,400 x 512 -> 256 at the bench size), forward and both backward products on
    the library's OWN GEMM kernels (round 6: neo_linear_forward / neo_linear_input_grad = k_sgemm, neo_linear_weight_grad = k_dw;
    rounds 4-5 called torch's matmul here - hipBLASLt `Cijk_*` kernels, 11.5 % of a training step in
    profiles/r05_train_step_kernel_stats.csv)."""

    @staticmethod
    def apply(lib_ctx, x, w):
        return _Linear.apply(lib_ctx, x, w, None, False)


def weight_grad(gy, x, bias=False, ctx=None):
    """dW (M, N) = gy^T x over the rows of gy (K, M) and x (K, N) (and db = column sums of gy with bias=True): the weight gradient of
    a linear layer, neo_linear_weight_grad."""
    gy, x = f32(gy, "gy"), f32(x, "x")
    if gy.dim() != 2 or x.dim() != 2 or gy.shape[0] != x.shape[0]:
        raise ValueError("gy (K, M) and x (K, N) must share their rows, got %s and %s" % (tuple(gy.shape), tuple(x.shape)))
    c = _ctx(gy, ctx)
    K, M, N = gy.shape[0], gy.shape[1], x.shape[1]
    gw = torch.zeros(M, N, device=gy.device, dtype=torch.float32)
    gb = torch.zeros(M, device=gy.device, dtype=torch.float32) if bias else None
    _lib.check(c.lib.neo_linear_weight_grad(c.handle, M, N, K, ptr(gy), M, ptr(x), N, ptr(gw), N, ptr(gb) if bias else None, c.stream()))
    return (gw, gb) if bias else gw


def project_latent(mlp, latent_cl, ctx=None):
    """G = F [W0_loc | W3_loc]^T per texel: latent_cl (texels, 512) channels-last -> (texels, 256), library GEMMs under autograd
    (the backward IS the gradient of the two local weight blocks - formed by neo_linear_weight_grad - and of the latent).
    W0_loc = pts_linears.0's columns of the 512 latent features, W3_loc = the same columns of the skip half of pts_linears.3
    (neo360/model.py:123-137)."""
    pe = mlp.input_ch * 21
    w0, w3 = mlp.pts_linears[0].weight, mlp.pts_linears[3].weight
    wcat = torch.cat([w0[:, pe:pe + 512], w3[:, 128 + pe:128 + pe + 512]], dim=0)          # (256, 512)
    return _LinearNoBias.apply(ctx, latent_cl, wcat)


def project_latent_all(mlps, latent_cl, ctx=None):
    """project_latent for several MLPs in ONE GEMM each way (round 6): G = F [W0_loc | W3_loc]_mlp0..^T, (texels, 256 len(mlps)); MLP i
    looks up columns [256 i, 256 i + 256) (gather_map(..., col, width, shared)).  Against one projection per MLP the latent is read
    once instead of len(mlps) times, its gradient is written once instead of being summed from len(mlps) full-size tensors, and the
    lookups' backward passes scatter into one buffer.  Returns (map, shared)."""
    blocks = []
    for mlp in mlps:
        pe = mlp.input_ch * 21
        w0, w3 = mlp.pts_linears[0].weight, mlp.pts_linears[3].weight
        blocks += [w0[:, pe:pe + 512], w3[:, 128 + pe:128 + pe + 512]]
    shared = {"buf": None}
    return _GradSink.apply(_LinearNoBias.apply(ctx, latent_cl, torch.cat(blocks, dim=0)), shared), shared


def nerfpp_mlp_projected(mlp, x_enc, cond_rows, world_feat, pre, nv, ctx=None):
    """nerfpp_mlp with the gathered projected latent `pre` (NV*P, 256) in place of the (NV*P, 512) local features: the
    512-wide segments of layer 0 and of the skip layer - forward, dX and dW - are not executed per row."""
    layers = mlp.ordered_layers()
    params = [l.weight for l in layers] + [l.bias for l in layers]
    return _TrainMLPPre.apply(ctx, mlp.input_ch, nv, x_enc, cond_rows, world_feat, pre, *params)


class _TrainMLP(torch.autograd.Function):
    """NeRFPPMLP on materialised rows with a native backward (neo_tp_mlp_train_forward / _backward)."""

    @staticmethod
    def forward(ctx_, lib_ctx, input_ch, nv, x_enc, cond_rows, world_feat, local_feat, *params):
        ws, bs = params[:9], params[9:]
        # the library reads rows of fixed widths: anything else is caught here, as the reference's matmuls would
        pe, npts = input_ch * 21, x_enc.shape[1]
        if x_enc.dim() != 3 or tuple(x_enc.shape) != (nv, npts, pe):
            raise ValueError("x_enc must be (NV, P, %d), got %s" % (pe, tuple(x_enc.shape)))
        for name, t, width in (("cond_rows", cond_rows, 27), ("world_feat", world_feat, 128), ("local_feat", local_feat, 512)):
            if tuple(t.shape) != (nv * npts, width):
                raise ValueError("%s must be (NV*P, %d) = (%d, %d), got %s" % (name, width, nv * npts, width, tuple(t.shape)))
        want = [(128, pe + 640), (128, 128), (128, 128), (128, 128 + pe + 640), (64, 155), (64, 64), (128, 128), (1, 128), (3, 64)]
        for i, (w, b) in enumerate(zip(ws, bs)):
            if tuple(w.shape) != want[i] or tuple(b.shape) != (want[i][0],):
                raise ValueError("NeRFPPMLP layer %d: weight %s / bias %s, expected %s / (%d,)"
                                 % (i, tuple(w.shape), tuple(b.shape), want[i], want[i][0]))
        # the three input tensors go to the library as they are: the (NV*P, 703) concatenation the reference forms is never built
        xe = f32(x_enc, "x_enc").reshape(-1, pe)
        lf, wf = f32(local_feat, "local_feat"), f32(world_feat, "world_feat")
        cond = f32(cond_rows, "cond_rows")
        c = _ctx(xe, lib_ctx)
        wd = [f32(w.detach(), "weight") for w in ws]
        bd = [f32(b.detach(), "bias") for b in bs]
        tab = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        raw_rgb = torch.empty(npts, 3, device=xe.device)
        raw_sigma = torch.empty(npts, 1, device=xe.device)
        # the activations live in a tensor owned by THIS autograd node: a training step runs several MLP forwards
        # (inside / outside the sphere, coarse / fine) before the first backward
        tape = torch.empty(c.lib.neo_tp_mlp_train_tape_floats(nv, npts), device=xe.device)
        _lib.check(c.lib.neo_tp_mlp_train_forward(c.handle, input_ch, tab(wd), tab(bd), ptr(xe), ptr(lf), ptr(wf), ptr(cond), nv,
                                                  npts, ptr(tape), ptr(raw_rgb), ptr(raw_sigma), c.stream()))
        ctx_.save_for_backward(xe, lf, wf, cond, tape, *wd)
        ctx_.meta = (c, input_ch, nv, npts, x_enc.shape, [tuple(w.shape) for w in ws], [tuple(b.shape) for b in bs])
        return raw_rgb, raw_sigma

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx_, g_rgb, g_sigma):
        xe, lf, wf, cond, tape, *wd = ctx_.saved_tensors
        c, input_ch, nv, npts, xshape, wshapes, bshapes = ctx_.meta
        if ctx_.needs_input_grad[4]:
            raise NotImplementedError("nerfpp_mlp: no gradient for cond_rows (the reference's view directions are data)")
        dev = xe.device
        g_rgb = f32(g_rgb.contiguous(), "g_rgb") if g_rgb is not None else torch.zeros(npts, 3, device=dev)
        g_sigma = f32(g_sigma.contiguous(), "g_sigma") if g_sigma is not None else torch.zeros(npts, 1, device=dev)
        gw = [torch.zeros(s, device=dev) for s in wshapes]
        gb = [torch.zeros(s, device=dev) for s in bshapes]
        # only the input gradients somebody asked for are computed (each its own dense tensor, no slicing afterwards)
        gx = torch.empty_like(xe) if ctx_.needs_input_grad[3] else None
        gworld = torch.empty_like(wf) if ctx_.needs_input_grad[5] else None
        glocal = torch.empty_like(lf) if ctx_.needs_input_grad[6] else None
        tab = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        _lib.check(c.lib.neo_tp_mlp_train_backward(c.handle, input_ch, tab(wd), ptr(xe), ptr(lf), ptr(wf), ptr(cond), nv, npts,
                                                   ptr(tape), ptr(g_rgb), ptr(g_sigma), tab(gw), tab(gb), ptr(gx), ptr(glocal),
                                                   ptr(gworld), c.stream()))
        return (None, None, None, gx.reshape(xshape) if gx is not None else None, None, gworld, glocal, *gw, *gb)


def nerfpp_mlp(mlp, x_enc, cond_rows, world_feat, local_feat, nv, ctx=None):
    """The reference's NeRFPPMLP.forward (neo360/model.py:110-158) with autograd support, for training steps:
    x_enc (NV,P,63|84) encoded camera-frame points, cond_rows (NV*P,27) view-direction encodings, world_feat (NV*P,128),
    local_feat (NV*P,512), view-major rows -> raw_rgb (P,3), raw_sigma (P,1) (pre-activation).  `mlp` is a
    models.NeRFPPMLP (parameter container, reference state_dict layout); gradients flow to all of its parameters and to
    x_enc / world_feat / local_feat (the latter two continue into gather_features' backward).  Exact fp32 matrix
    arithmetic; the fused inference evaluators are untouched."""
    layers = mlp.ordered_layers()
    params = [l.weight for l in layers] + [l.bias for l in layers]
    return _TrainMLP.apply(ctx, mlp.input_ch, nv, x_enc, cond_rows, world_feat, local_feat, *params)


class _TrainVanillaMLP(torch.autograd.Function):
    """Vanilla NeRFMLP on materialised rows with a native backward (neo_vanilla_mlp_train_forward / _backward)."""

    @staticmethod
    def forward(ctx_, lib_ctx, x_enc, dir_enc, *params):
        ws, bs = params[:12], params[12:]
        if x_enc.dim() != 3 or x_enc.shape[-1] != 63 or tuple(dir_enc.shape) != (x_enc.shape[0], 27):
            raise ValueError("x_enc must be (B, N, 63) and dir_enc (B, 27), got %s / %s" % (tuple(x_enc.shape), tuple(dir_enc.shape)))
        want = [(256, 63)] + [(256, 256)] * 4 + [(256, 319)] + [(256, 256)] * 2 + [(128, 283), (256, 256), (1, 256), (3, 128)]
        for i, (w, b) in enumerate(zip(ws, bs)):
            if tuple(w.shape) != want[i] or tuple(b.shape) != (want[i][0],):
                raise ValueError("NeRFMLP layer %d: weight %s / bias %s, expected %s / (%d,)"
                                 % (i, tuple(w.shape), tuple(b.shape), want[i], want[i][0]))
        B, N, F = x_enc.shape
        x0 = f32(x_enc, "x_enc").reshape(-1, F).contiguous()
        cond = torch.tile(f32(dir_enc, "dir_enc")[:, None, :], (1, N, 1)).reshape(-1, dir_enc.shape[-1]).contiguous()
        c = _ctx(x0, lib_ctx)
        wd = [f32(w.detach(), "weight") for w in ws]
        bd = [f32(b.detach(), "bias") for b in bs]
        tab = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        R = B * N
        raw_rgb, raw_sigma = torch.empty(R, 3, device=x0.device), torch.empty(R, 1, device=x0.device)
        tape = torch.empty(c.lib.neo_vanilla_mlp_train_tape_floats(R), device=x0.device)
        _lib.check(c.lib.neo_vanilla_mlp_train_forward(c.handle, tab(wd), tab(bd), ptr(x0), ptr(cond), R, ptr(tape), ptr(raw_rgb),
                                                       ptr(raw_sigma), c.stream()))
        ctx_.save_for_backward(x0, cond, tape, *wd)
        ctx_.meta = (c, B, N, F, [tuple(w.shape) for w in ws], [tuple(b.shape) for b in bs])
        return raw_rgb.reshape(B, N, 3), raw_sigma.reshape(B, N, 1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx_, g_rgb, g_sigma):
        x0, cond, tape, *wd = ctx_.saved_tensors
        c, B, N, F, wshapes, bshapes = ctx_.meta
        dev, R = x0.device, B * N
        g_rgb = f32(g_rgb.reshape(R, 3).contiguous(), "g_rgb") if g_rgb is not None else torch.zeros(R, 3, device=dev)
        g_sigma = f32(g_sigma.reshape(R, 1).contiguous(), "g_sigma") if g_sigma is not None else torch.zeros(R, 1, device=dev)
        gw = [torch.zeros(s, device=dev) for s in wshapes]
        gb = [torch.zeros(s, device=dev) for s in bshapes]
        g_x0 = torch.empty_like(x0) if ctx_.needs_input_grad[1] else None
        g_cond = torch.empty_like(cond) if ctx_.needs_input_grad[2] else None
        tab = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        _lib.check(c.lib.neo_vanilla_mlp_train_backward(c.handle, tab(wd), ptr(x0), ptr(cond), R, ptr(tape), ptr(g_rgb), ptr(g_sigma),
                                                        tab(gw), tab(gb), ptr(g_x0), ptr(g_cond), c.stream()))
        gx = g_x0.reshape(B, N, F) if g_x0 is not None else None
        gd = g_cond.reshape(B, N, -1).sum(dim=1) if g_cond is not None else None      # the tiling's backward
        return (None, gx, gd, *gw, *gb)


def nerf_mlp(mlp, x_enc, dir_enc, ctx=None):
    """The reference's vanilla NeRFMLP.forward (vanilla_nerf/model.py:100-125) with autograd support: x_enc (B,N,63) encoded
    sample points, dir_enc (B,27) per-ray view-direction encodings -> raw_rgb (B,N,3), raw_sigma (B,N,1).  `mlp` is a
    models.NeRFMLP; gradients flow to all 24 parameter tensors and to both inputs.  Exact fp32 matrix arithmetic."""
    layers = mlp.ordered_layers()
    params = [l.weight for l in layers] + [l.bias for l in layers]
    return _TrainVanillaMLP.apply(ctx, x_enc, dir_enc, *params)


# ---- the module-level training call (neo360/model.py:697-820 calls self.model(batch, randomized=True, ...)) ------------------

def _draws(module, B, randomized, seed, c, n0, n1, noise):
    """Every uniform a randomized call consumes, for ALL B rows of the call, from the library's counter-based generator:
    stream ids 0 / 1 level-0 jitter inside / outside the sphere, 2 / 3 level-1 quantiles, 4..7 the density noise
    (neo360/model.py:381-384: torch.rand_like(raw_sigma) * density_noise) of (level, region).  Row r of every table belongs
    to ray r of the call whatever the chunking, which is what the fused neo_tp_render_train draws (rows 0..R-1 from one seed)."""
    if not randomized:
        return None, 0
    seed = int(seed) if seed is not None else int(torch.randint(1, 2 ** 62, (1,)).item())
    seed = seed or 1
    d = dict(u_fg=rand_uniform(seed, 0, B, n0 + 1, ctx=c), u_bg=rand_uniform(seed, 1, B, n0 + 1, ctx=c),
             q_fg=rand_uniform(seed, 2, B, n1, ctx=c), q_bg=rand_uniform(seed, 3, B, n1, ctx=c))
    if noise:
        for level, n in ((0, n0 + 1), (1, n0 + 1 + n1)):
            d["n_fg%d" % level] = rand_uniform(seed, 4 + 2 * level, B, n, ctx=c)
            d["n_bg%d" % level] = rand_uniform(seed, 5 + 2 * level, B, n, ctx=c)
    return d, seed


def tp_render_train(module, rays, randomized, white_bkgd, maps, chunk=None, seed=None):
    """NeRF_TP.forward(out_depth=False) WITH autograd: per level (comp_rgb, fg_weights, bg_weights, fg_sdist, bg_sdist,
    bg_acc) exactly as the fused call returns them (neo360/model.py:531-579), built from the differentiable operators of
    this file - level-0 samples (sample_level0), lookups (gather_features: gradients reach the four feature maps and,
    through them, an encoder), encodings, NeRFPPMLP with its native backward (nerfpp_mlp: gradients reach all four
    MLPs' parameters), the reference's activations, compositing (composite), hierarchical resampling on detached weights
    (helper.py:224) - so the reference's training_step (model.py:697-820: rgb loss on both levels + eff_distloss on the
    weights) runs unchanged on `loss.backward()`.  maps = (plane_xz, plane_xy, plane_yz, latent) NCHW tensors (the
    encoder's outputs, or the tensors given to set_scene).  randomized draws come from the library's counter-based
    generator with the stream ids of the fused call (0 / 1 level-0 jitter fg / bg, 2 / 3 level-1 quantiles), ONE table per
    stream for all rays of the call, sliced per chunk: both paths see the same samples for one seed at any `chunk`
    (round 5; the first version re-seeded every chunk).  `module.density_noise` (model.py:381-384, uniform noise on the raw
    density when randomized) is drawn from streams 4..7.  All rays form ONE reference chunk unless `chunk` is given."""
    rays_o = f32(rays["rays_o"], "rays_o")
    B = rays_o.shape[0]
    c = module._context(rays_o.device)
    n0, n1 = module.num_coarse_samples, module.num_fine_samples
    with torch.no_grad():
        draws, _ = _draws(module, B, randomized, seed, c, n0, n1, module.density_noise != 0.0)
    step = int(chunk) if chunk is not None and int(chunk) < B else B
    # projected-space path (round 5, default): the latent goes through [W0_loc | W3_loc] of each MLP ONCE per call, in texel space;
    # the per-row work gathers 256 instead of 512 channels and skips the 512-wide GEMM segments (forward, dX, dW)
    projected = getattr(module, "train_projected", None)
    if projected is None:
        projected = os.environ.get("NEO360_TRAIN_PROJECTED", "1") != "0"
    proj = {"latent_cl": channels_last_rows(maps[3], ctx=c), "plane_grads": {}} if projected else None
    parts = []
    for i in range(0, B, max(step, 1)):
        sub = {k: (v[i:i + step] if k in ("rays_o", "rays_d", "viewdirs") else v) for k, v in rays.items()}
        rows = {k: v[i:i + step] for k, v in draws.items()} if draws is not None else None
        parts.append(_tp_render_train_chunk(module, sub, randomized, white_bkgd, maps, rows, proj))
    if len(parts) == 1:
        return parts[0]
    return [tuple(torch.cat([p[lv][j] for p in parts], dim=0) for j in range(6)) for lv in range(2)]


def _tp_render_train_chunk(module, rays, randomized, white_bkgd, maps, draws, proj=None):
    """One reference chunk of tp_render_train; `draws` = this chunk's rows of the call's uniform tables (None: deterministic);
    `proj` = the call's cache of projected maps (None: the local features are gathered and multiplied per row)."""
    from . import ops
    rays_o, rays_d, viewdirs = f32(rays["rays_o"], "rays_o"), f32(rays["rays_d"], "rays_d"), f32(rays["viewdirs"], "viewdirs")
    B = rays_o.shape[0]
    dev = rays_o.device
    c = module._context(dev)
    poses = f32(rays["src_poses"], "src_poses")
    NV = poses.shape[0]
    n0, n1 = module.num_coarse_samples, module.num_fine_samples
    with torch.no_grad():
        far, _ = ops.intersect_sphere(rays_o, rays_d, ctx=c)                                   # model.py:278
        u_fg, u_bg = (draws["u_fg"], draws["u_bg"]) if draws is not None else (None, None)
        fg_t, bg_s = sample_level0(far, n0, u_fg, u_bg, ctx=c)
        rot = poses[:, :3, :3].transpose(1, 2)
        dir_cam = (rot[:, None, :, :] * viewdirs[None, :, None, :]).sum(-1)                    # (NV,B,3) = R_v^T d: model.py:339-341 (3-term sums, no library GEMM)
        d_enc = ops.pos_enc(dir_cam, 0, module.deg_view, ctx=c)                                # (NV,B,27)
    mlps = module._mlps()
    out = []
    for level in range(2):
        N = fg_t.shape[1]
        with torch.no_grad():
            cond = d_enc.repeat(1, N, 1).reshape(-1, d_enc.shape[-1])      # row (v, j) carries ray j mod B (model.py:357-360, quirk Q1)
            # lookup points + per-view encodings of both regions: ONE kernel each, the evaluators' own point set-up (round 5;
            # the first version restated helper.py:401-451 / util.py:52-70 in eager torch)
            fg_p, fg_x = train_points(module, 0, rays_o, rays_d, fg_t, None, poses, ctx=c)
            bg_lin, bg_x = train_points(module, 1, rays_o, rays_d, bg_s, far, poses, ctx=c)
        res = {}
        for name, mlp, look, x_enc in (("fg", mlps[level], fg_p, fg_x), ("bg", mlps[2 + level], bg_lin, bg_x)):
            if proj is not None:
                if "G" not in proj:                                     # (texels, 4 x 256): one texel-space GEMM for the four MLPs, once per call
                    proj["G"], proj["shared"] = project_latent_all(mlps, proj["latent_cl"], ctx=c)
                share = getattr(module, "train_shared_grads", True)     # False: one private gradient buffer per lookup (slower, no sharing)
                world = gather_planes(module, look, maps[0], maps[1], maps[2], maps[3], rays, shared=proj["plane_grads"] if share else None)
                slot = level + (2 if name == "bg" else 0)              # mlps = (fg coarse, fg fine, bg coarse, bg fine)
                pre = gather_map(module, proj["G"], look, rays, col=256 * slot, width=256, shared=proj["shared"] if share else None)
                raw_rgb, raw_sigma = nerfpp_mlp_projected(mlp, x_enc, cond, world, pre, NV, ctx=c)
            else:
                world, local = gather_features(module, look, maps[0], maps[1], maps[2], maps[3], rays)
                raw_rgb, raw_sigma = nerfpp_mlp(mlp, x_enc, cond, world, local, NV, ctx=c)
            noisy = draws is not None and module.density_noise != 0.0                            # model.py:381-384
            res[name] = activate(raw_rgb, raw_sigma, draws["n_%s%d" % (name, level)] if noisy else None,
                                 float(module.density_noise) if noisy else 0.0, ctx=c).reshape(B, N, 4)   # model.py:380-385
        fg_c, _, fg_w, lam, _ = composite(1, res["fg"], None, fg_t, rays_d, far, white_bkgd, ctx=c)
        bg_c, bg_acc, bg_w, _, _ = composite(2, res["bg"], None, bg_s, None, None, white_bkgd, ctx=c)
        rgb = fg_c + lam * bg_c
        fg_sd = 0.5 * (fg_t[..., 1:] + fg_t[..., :-1])                                          # model.py:564-571
        fg_sd = torch.cat([fg_sd, (fg_sd[:, -1] + (fg_sd[:, -1] - fg_sd[:, -2])).unsqueeze(-1)], dim=-1)
        bg_sd = torch.cat([0.5 * (bg_s[..., 1:] + bg_s[..., :-1]), bg_s[..., -1:]], dim=-1)
        out.append((rgb, fg_w, bg_w, fg_sd, bg_sd, bg_acc))
        if level == 0:
            with torch.no_grad():
                if draws is not None:
                    fg_t = resample_u(fg_t, fg_w, draws["q_fg"], False, ctx=c)
                    bg_s = resample_u(bg_s, bg_w, draws["q_bg"], True, ctx=c)
                else:
                    fg_t = ops.resample(fg_t, fg_w.detach(), n1, False, ctx=c)
                    bg_s = ops.resample(bg_s, bg_w.detach(), n1, True, ctx=c)
    flags = c.poll_flags()
    module._raise_flags(flags)
    return out


# ---- the vanilla renderer's training call (vanilla_nerf/model.py:281-283 calls self.model(batch, randomized=True, ...)) ---------

def nerf_render_train(module, rays, randomized, white_bkgd, near, far, seed=None, return_samples=False):
    """models.NeRF.forward WITH autograd / stratified sampling (vanilla_nerf/model.py:154-216 under training_step :255-300):
    per level (comp_rgb (B,3), acc (B,), depth (B,)).  Chained from the operators of this file: level-0 samples along
    `viewdirs` between the scalar near / far (helper.py:415-442; randomized = one uniform per sample inside its stratum), 63-d
    encodings (neo_pos_enc), the vanilla NeRFMLP with its native backward (nerf_mlp: gradients reach all 24 parameter tensors of
    each MLP), `noise_std` (model.py:194-195: uniform noise on the raw density when randomized), the reference's activations,
    compositing with `rays_d` norms (composite mode 0), inverse-CDF resampling on detached weights (helper.py:610-616).
    Uniforms: the library's counter-based generator, stream 0 = level-0 jitter, 2 = level-1 quantiles, 4 / 6 = density noise of
    level 0 / 1 (torch.manual_seed makes the default seed repeatable).  return_samples: also return [t0 (B,n0+1), t1 (B,n0+1+n1)],
    the sample positions of the two levels (tests evaluate the oracle at them)."""
    from . import ops
    rays_o, rays_d, viewdirs = f32(rays["rays_o"], "rays_o"), f32(rays["rays_d"], "rays_d"), f32(rays["viewdirs"], "viewdirs")
    B = rays_o.shape[0]
    dev = rays_o.device
    c = module._context(dev)
    n0, n1 = module.num_coarse_samples, module.num_fine_samples
    noise = float(module.noise_std) if randomized else 0.0
    used = []
    with torch.no_grad():
        if randomized:
            seed = int(seed) if seed is not None else int(torch.randint(1, 2 ** 62, (1,)).item())
            seed = seed or 1
        # helper.py:424-429: linspace(0,1,n+1) -> near (1 - t) + far t, fp32 (the CPU linspace the library's tables use)
        lin = torch.linspace(0.0, 1.0, n0 + 1).to(dev)
        edges = float(near) * (1.0 - lin) + float(far) * lin
        if randomized:                                                                          # helper.py:431-436
            mids = 0.5 * (edges[1:] + edges[:-1])
            upper, lower = torch.cat([mids, edges[-1:]]), torch.cat([edges[:1], mids])
            t = lower + (upper - lower) * rand_uniform(seed, 0, B, n0 + 1, ctx=c)
        else:
            t = edges[None, :].expand(B, n0 + 1).contiguous()
        d_enc = ops.pos_enc(viewdirs, 0, module.deg_view, ctx=c)                                # (B,27), model.py:190
    out = []
    for level, mlp in enumerate((module.coarse_mlp, module.fine_mlp)):
        N = t.shape[1]
        used.append(t)
        with torch.no_grad():
            pts = rays_o[:, None, :] + t[..., None] * viewdirs[:, None, :]                      # cast_rays along viewdirs (:161, :177)
            x_enc = ops.pos_enc(pts, module.min_deg_point, module.max_deg_point, ctx=c)         # (B,N,63)
        raw_rgb, raw_sigma = nerf_mlp(mlp, x_enc, d_enc, ctx=c)
        u_noise = rand_uniform(seed, 4 + 2 * level, B, N, ctx=c) if noise > 0.0 else None       # model.py:194-195
        packed = activate(raw_rgb, raw_sigma, u_noise, noise, ctx=c).reshape(B, N, 4)           # model.py:200-205
        comp_rgb, acc, w, _, depth = composite(0, packed, None, t, rays_d, None, white_bkgd, ctx=c)
        out.append((comp_rgb, acc, depth))
        if level == 0:
            with torch.no_grad():                                                               # helper.py:610-616 (detached)
                if randomized:
                    t = resample_u(t, w, rand_uniform(seed, 2, B, n1, ctx=c), False, ctx=c)
                else:
                    t = ops.resample(t, w.detach(), n1, False, ctx=c)
    module._raise_flags(c.poll_flags())
    return (out, used) if return_samples else out


# ---- PixelNeRF decoder: the training call (vanilla_nerf/model_pixel.py:174-258 under the LitPixelNeRF training_step) ------------
class _PixTrainMLPPre(torch.autograd.Function):
    """PixelNeRF's MLP on the projected latent as ONE native chain each way (neo_pix_mlp_train_forward_pre / _backward_pre):
    activations go layer to layer inside the library, the backward returns all 18 parameter gradients, dL/dpre and - if asked -
    dL/dx_enc.  The gradient of W0's latent columns flows through `pre` into the texel-space GEMM (project_pixel_latent)."""

    @staticmethod
    def forward(ctx_, lib_ctx, nv, x_enc, cond_rows, pre, *params):
        ws, bs = params[:9], params[9:]
        npts = x_enc.shape[1]
        if x_enc.dim() != 3 or tuple(x_enc.shape) != (nv, npts, 63):
            raise ValueError("x_enc must be (NV, P, 63), got %s" % (tuple(x_enc.shape),))
        for name, t, width in (("cond_rows", cond_rows, 27), ("pre", pre, 128)):
            if tuple(t.shape) != (nv * npts, width):
                raise ValueError("%s must be (NV*P, %d) = (%d, %d), got %s" % (name, width, nv * npts, width, tuple(t.shape)))
        want = [(128, 575), (128, 128), (128, 128), (128, 128), (128, 155), (128, 128), (128, 128), (1, 128), (3, 128)]
        for i, (w, b) in enumerate(zip(ws, bs)):
            if tuple(w.shape) != want[i] or tuple(b.shape) != (want[i][0],):
                raise ValueError("PixelNeRF MLP layer %d: weight %s / bias %s, expected %s / (%d,)"
                                 % (i, tuple(w.shape), tuple(b.shape), want[i], want[i][0]))
        xe = f32(x_enc, "x_enc").reshape(-1, 63)
        pf, cond = f32(pre, "pre"), f32(cond_rows, "cond_rows")
        c = _ctx(xe, lib_ctx)
        wd = [f32(w.detach(), "weight") for w in ws]
        bd = [f32(b.detach(), "bias") for b in bs]
        tab = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        raw_rgb = torch.empty(npts, 3, device=xe.device)
        raw_sigma = torch.empty(npts, 1, device=xe.device)
        tape = torch.empty(c.lib.neo_pix_mlp_train_tape_floats(nv, npts), device=xe.device)
        _lib.check(c.lib.neo_pix_mlp_train_forward_pre(c.handle, tab(wd), tab(bd), ptr(xe), ptr(pf), ptr(cond), nv, npts, ptr(tape),
                                                       ptr(raw_rgb), ptr(raw_sigma), c.stream()))
        ctx_.save_for_backward(xe, cond, tape, *wd)
        ctx_.meta = (c, nv, npts, x_enc.shape, [tuple(w.shape) for w in ws], [tuple(b.shape) for b in bs])
        return raw_rgb, raw_sigma

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx_, g_rgb, g_sigma):
        xe, cond, tape, *wd = ctx_.saved_tensors
        c, nv, npts, xshape, wshapes, bshapes = ctx_.meta
        if ctx_.needs_input_grad[3]:
            raise NotImplementedError("pixel_mlp: no gradient for cond_rows (the reference's view directions are data)")
        dev = xe.device
        g_rgb = f32(g_rgb.contiguous(), "g_rgb") if g_rgb is not None else torch.zeros(npts, 3, device=dev)
        g_sigma = f32(g_sigma.contiguous(), "g_sigma") if g_sigma is not None else torch.zeros(npts, 1, device=dev)
        gw = [torch.zeros(s, device=dev) for s in wshapes]
        gb = [torch.zeros(s, device=dev) for s in bshapes]
        gx = torch.empty_like(xe) if ctx_.needs_input_grad[2] else None
        gpre = torch.empty(nv * npts, 128, device=dev)
        tab = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        _lib.check(c.lib.neo_pix_mlp_train_backward_pre(c.handle, tab(wd), ptr(xe), ptr(cond), nv, npts, ptr(tape), ptr(g_rgb), ptr(g_sigma),
                                                        tab(gw), tab(gb), ptr(gx), ptr(gpre), c.stream()))
        return (None, None, gx.reshape(xshape) if gx is not None else None, None, gpre, *gw, *gb)


def pixel_mlp_fused(mlp, x_enc, cond_rows, pre, nv, ctx=None):
    """pixel_mlp_projected as one native chain each way (round 6; the analogue of nerfpp_mlp_projected for the PixelNeRF decoder):
    same inputs, same outputs, gradients to all 18 parameter tensors and to `pre`."""
    layers = mlp.ordered_layers()
    params = [l.weight for l in layers] + [l.bias for l in layers]
    return _PixTrainMLPPre.apply(ctx, nv, x_enc, cond_rows, pre, *params)


def pixel_mlp_projected(mlp, x_enc, cond_rows, pre, nv, ctx=None):
    """PixelNeRF's late-fusion MLP (model_pixel.py:96-131) under autograd, every product on the library's exact-fp32 GEMMs
    (`linear`): x_enc (NV,P,63), cond_rows (NV*P,27), pre (NV*P,128) = the gathered latent projected through the local columns
    of pts_linears.0 (W0 [:, 63:575]; the 512-wide product is formed per texel by project_pixel_latent) -> raw_rgb (P,3),
    raw_sigma (P,1).  The view means, the concatenation [bottleneck | cond] (two products into one sum) and the ReLU of
    the first layer are torch elementwise ops between the GEMMs."""
    P = x_enc.shape[1]
    xe = x_enc.reshape(-1, x_enc.shape[-1])
    L = mlp.pts_linears
    h = torch.relu(linear(xe, L[0].weight[:, :63], L[0].bias, ctx=ctx) + pre)
    for i in (1, 2, 3):
        h = linear(h, L[i].weight, L[i].bias, relu=True, ctx=ctx)
    bott = linear(h, mlp.bottleneck_layer.weight, mlp.bottleneck_layer.bias, ctx=ctx)                 # per view, before the mean (:113-114)
    hm = h.reshape(nv, P, -1).mean(0)                                                                  # combine_interleaved "average"
    raw_sigma = linear(hm, mlp.density_layer.weight, mlp.density_layer.bias, ctx=ctx)
    v0 = mlp.views_linear[0]
    y = linear(bott, v0.weight[:, :128], v0.bias, ctx=ctx) + linear(cond_rows, v0.weight[:, 128:], None, ctx=ctx)
    y = torch.relu(y.reshape(nv, P, -1).mean(0))
    y = linear(y, mlp.views_linear[1].weight, mlp.views_linear[1].bias, relu=True, ctx=ctx)
    return linear(y, mlp.rgb_layer.weight, mlp.rgb_layer.bias, ctx=ctx), raw_sigma


def project_pixel_latent(mlp, latent_cl, ctx=None):
    """G = F W0_loc^T per texel (texels, 512) -> (texels, 128): the latent's only path into PixelNeRF's MLP (no skip fires at
    depth 4), as project_latent does for NeRFPPMLP."""
    return _LinearNoBias.apply(ctx, latent_cl, mlp.pts_linears[0].weight[:, 63:575])


def pix_render_train(module, rays, randomized, white_bkgd, near, far, latent, seed=None, return_samples=False):
    """models.PixelNeRF.forward WITH autograd / stratified sampling (model_pixel.py:174-258): per level (comp_rgb (B,3), acc (B,),
    depth (B,)).  Level-0 samples along `rays_d` between the scalar near / far (helper.py:415-442), lookup points and camera-frame
    encodings by neo_tp_train_points, the latent projected per texel and gathered at the decoder's taps (gather_map kind "pix":
    gradients reach `latent` (NV,512,Hf,Wf) - an attached encoder's output - and the local weight columns), the MLP on the
    library's GEMMs (pixel_mlp_projected), `noise_std` (:235-236: uniform noise on the raw density when randomized), sigmoid /
    ReLU (:238-239), vanilla compositing (composite mode 0), inverse-CDF resampling on detached weights.  Uniform streams as
    nerf_render_train: 0 level-0 jitter, 2 level-1 quantiles, 4 / 6 density noise.  The direction encodings are tiled with the
    reference's (1, N, 1) pattern on a (NV, B, 27) tensor (:219-222: row (v, j) carries ray j mod B)."""
    from . import ops
    rays_o, rays_d, viewdirs = f32(rays["rays_o"], "rays_o"), f32(rays["rays_d"], "rays_d"), f32(rays["viewdirs"], "viewdirs")
    B = rays_o.shape[0]
    dev = rays_o.device
    c = module._context(dev)
    n0, n1 = module.num_coarse_samples, module.num_fine_samples
    noise = float(module.noise_std) if randomized else 0.0
    poses = f32(rays["src_poses"], "src_poses")
    NV = poses.shape[0]
    used = []
    with torch.no_grad():
        if randomized:
            seed = int(seed) if seed is not None else int(torch.randint(1, 2 ** 62, (1,)).item())
            seed = seed or 1
        lin = torch.linspace(0.0, 1.0, n0 + 1).to(dev)
        edges = float(near) * (1.0 - lin) + float(far) * lin
        if randomized:
            mids = 0.5 * (edges[1:] + edges[:-1])
            upper, lower = torch.cat([mids, edges[-1:]]), torch.cat([edges[:1], mids])
            t = lower + (upper - lower) * rand_uniform(seed, 0, B, n0 + 1, ctx=c)
        else:
            t = edges[None, :].expand(B, n0 + 1).contiguous()
        rot = poses[:, :3, :3].transpose(1, 2)                                                  # util.py:45-49: R^T d per view
        dir_cam = (rot[:, None, :, :] * viewdirs[None, :, None, :]).sum(-1)                     # (NV,B,3), as tp_render_train
        d_enc = ops.pos_enc(dir_cam, 0, 4, ctx=c)                                               # (NV,B,27)
    latent_cl = channels_last_rows(latent, ctx=c)                                                # (NV Hf Wf, 512) channels-last, under autograd
    out = []
    for level, mlp in enumerate((module.coarse_mlp, module.fine_mlp)):
        N = t.shape[1]
        used.append(t)
        with torch.no_grad():
            cond = d_enc.repeat(1, N, 1).reshape(-1, d_enc.shape[-1])                           # tile (1,N,1) of (NV,1,B,27) (:219-222)
            look, x_enc = train_points(module, 0, rays_o, rays_d, t, None, poses, ctx=c)
        pre = gather_map(module, project_pixel_latent(mlp, latent_cl, ctx=c), look, rays, kind="pix")
        # the MLP as one native chain each way (round 6); `module.train_fused = False`: the per-layer operators (pixel_mlp_projected)
        mlp_fn = pixel_mlp_fused if getattr(module, "train_fused", True) else pixel_mlp_projected
        raw_rgb, raw_sigma = mlp_fn(mlp, x_enc, cond, pre, NV, ctx=c)
        raw_sigma = raw_sigma.reshape(B, N)
        if noise > 0.0:
            raw_sigma = raw_sigma + rand_uniform(seed, 4 + 2 * level, B, N, ctx=c) * noise
        rgb, sigma = torch.sigmoid(raw_rgb.reshape(B, N, 3)), torch.relu(raw_sigma)
        comp_rgb, acc, w, _, depth = composite(0, rgb, sigma, t, rays_d, None, white_bkgd, ctx=c)
        out.append((comp_rgb, acc, depth))
        if level == 0:
            with torch.no_grad():
                if randomized:
                    t = resample_u(t, w, rand_uniform(seed, 2, B, n1, ctx=c), False, ctx=c)
                else:
                    t = ops.resample(t, w.detach(), n1, False, ctx=c)
    module._raise_flags(c.poll_flags())
    return (out, used) if return_samples else out


# ---- Mip-NeRF 360: the training call (mipnerf360/model.py:236-365 under LitMipNeRF360.training_step :436-470) --------------------
_EPS32 = float(torch.finfo(torch.float32).eps)


def mip_resample_u(s_prev, w_prev, n, near, far, dilate, dilation, anneal, u, jitter=None, ctx=None):
    """One proposal-resampling step with the caller's quantile table u (n) and, when given, one jitter per ray (R,)
    (neo_mip_resample_u; helper.py:343-396).  Returns sdist, tdist (R, n+1); no gradients (stop_level_grad)."""
    s_prev, w_prev, u = f32(s_prev, "s_prev"), f32(w_prev, "w_prev"), f32(u, "u")
    c = _ctx(s_prev, ctx)
    R, n_prev = w_prev.shape
    sdist = torch.empty(R, n + 1, device=s_prev.device)
    tdist = torch.empty(R, n + 1, device=s_prev.device)
    jit = f32(jitter, "jitter").reshape(-1) if jitter is not None else None
    _lib.check(c.lib.neo_mip_resample_u(c.handle, ptr(s_prev), ptr(w_prev), R, n_prev, int(bool(dilate)), float(dilation),
                                        float(anneal), n, ptr(u), ptr(jit), float(near), float(far), ptr(sdist), ptr(tdist), c.stream()))
    return sdist, tdist


def mip_encode(rays_o, rays_d, radii, tdist, pos_basis_t, ctx=None):
    """(R n, 504) integrated positional encodings of the intervals of tdist (R, n+1) (neo_mip_encode; helper.py:33-88, 278-334)."""
    rays_o, rays_d, radii, tdist = f32(rays_o, "rays_o"), f32(rays_d, "rays_d"), f32(radii, "radii"), f32(tdist, "tdist")
    basis = f32(pos_basis_t, "pos_basis_t")
    c = _ctx(tdist, ctx)
    R, n1 = tdist.shape
    out = torch.empty(R * (n1 - 1), 504, device=tdist.device)
    _lib.check(c.lib.neo_mip_encode(c.handle, ptr(rays_o), ptr(rays_d), ptr(radii), ptr(tdist), ptr(basis), R, n1 - 1, ptr(out), c.stream()))
    return out


class _MipComposite(torch.autograd.Function):
    """compute_alpha_weights(opaque_background=True) + volumetric_rendering (helper.py:246-275) with a native backward
    (neo_mip_composite / neo_mip_composite_backward): (rgb (R,n,3), density (R,n)) -> weights (R,n), colour (R,3)."""

    @staticmethod
    def forward(ctx_, rgb, density, tdist, rays_d, bg, lib_ctx):
        rgbdens = torch.cat([f32(rgb, "rgb"), f32(density, "density")[..., None]], dim=-1).contiguous()
        tdist, rays_d = f32(tdist, "tdist"), f32(rays_d, "rays_d")
        c = _ctx(tdist, lib_ctx)
        R, n1 = tdist.shape
        w = torch.empty(R, n1 - 1, device=tdist.device)
        out = torch.empty(R, 3, device=tdist.device)
        _lib.check(c.lib.neo_mip_composite(c.handle, ptr(rgbdens), ptr(tdist), ptr(rays_d), R, n1 - 1, float(bg), ptr(w), ptr(out), c.stream()))
        ctx_.save_for_backward(rgbdens, tdist, rays_d)
        ctx_.meta = (c, float(bg))
        return w, out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx_, g_w, g_c):
        rgbdens, tdist, rays_d = ctx_.saved_tensors
        c, bg = ctx_.meta
        R, n1 = tdist.shape
        g = torch.empty(R, n1 - 1, 4, device=tdist.device)
        gw = f32(g_w.contiguous(), "g_w") if g_w is not None else None
        gc = f32(g_c.contiguous(), "g_c") if g_c is not None else None
        _lib.check(c.lib.neo_mip_composite_backward(c.handle, ptr(rgbdens), ptr(tdist), ptr(rays_d), R, n1 - 1, bg, ptr(gw), ptr(gc),
                                                    ptr(g), c.stream()))
        return g[..., :3], g[..., 3], None, None, None, None


def mip_composite(rgb, density, tdist, rays_d, bg=1.0, ctx=None):
    return _MipComposite.apply(rgb, density, tdist, rays_d, bg, ctx)


def mip_mlp(mlp, x0, d_enc, n, ctx=None):
    """MipNeRF360MLP.forward (model.py:107-176) on encoded rows x0 (R n, 504) under autograd, every product on the library's
    exact-fp32 GEMMs (`linear`): trunk of `netdepth` ReLU layers, the encoding concatenated again after layer 4 (two products into
    one sum instead of the concatenation), density = softplus(raw - 1); with the rgb branch: bottleneck, view layer on
    [bottleneck | dir_enc] (the direction term once per RAY, broadcast over its n intervals), rgb = sigmoid(.) (1 + 2 pad) - pad.
    d_enc (R, 27).  Returns density (R, n), rgb (R, n, 3) (zeros without the branch)."""
    W = mlp.netwidth
    h = x0
    for i, layer in enumerate(mlp.pts_linear):
        if i > 0 and (i - 1) % 4 == 0 and (i - 1) > 0:          # the layer after a skip concat: input [h | x0]
            h = torch.relu(linear(h, layer.weight[:, :W], layer.bias, ctx=ctx) + linear(x0, layer.weight[:, W:], None, ctx=ctx))
        else:
            h = linear(h, layer.weight, layer.bias, relu=True, ctx=ctx)
    # a skip after the LAST layer (depth 5, 9, ..) would widen the heads' input: not a shape the reference's defaults produce
    raw = linear(h, mlp.density_layer.weight, mlp.density_layer.bias, ctx=ctx)
    R = x0.shape[0] // n
    density = torch.nn.functional.softplus(raw.reshape(R, n) + (-1.0))
    if mlp.disable_rgb:
        return density, torch.zeros(R, n, 3, device=x0.device)
    bott = linear(h, mlp.bottleneck_layer.weight, mlp.bottleneck_layer.bias, ctx=ctx)
    v0 = mlp.views_linear[0]
    y = linear(bott, v0.weight[:, :256], v0.bias, ctx=ctx).reshape(R, n, -1) + linear(d_enc, v0.weight[:, 256:], None, ctx=ctx)[:, None, :]
    y = torch.relu(y).reshape(R * n, -1)
    rgb = torch.sigmoid(linear(y, mlp.rgb_layer.weight, mlp.rgb_layer.bias, ctx=ctx)).reshape(R, n, 3)
    return density, rgb * (1 + 2 * 0.001) - 0.001


class _MipTrainMLP(torch.autograd.Function):
    """MipNeRF360MLP.forward under autograd as ONE native chain each way (neo_mip_mlp_train_forward / _backward, round 6): trunk,
    skip layer, heads and both activations inside the library; the backward returns every parameter gradient.  x0 / d_enc are data
    (sdist is detached: stop_level_grad), so no input gradients."""

    @staticmethod
    def forward(ctx_, lib_ctx, width, depth, rgb, n, x0, d_enc, *params):
        nl = depth + (4 if rgb else 1)
        ws, bs = params[:nl], params[nl:]
        rows = x0.shape[0]
        if x0.dim() != 2 or x0.shape[1] != 504 or rows % n:
            raise ValueError("x0 must be (R n, 504) with n = %d, got %s" % (n, tuple(x0.shape)))
        R = rows // n
        if rgb and tuple(d_enc.shape) != (R, 27):
            raise ValueError("d_enc must be (R, 27) = (%d, 27), got %s" % (R, tuple(d_enc.shape)))
        want = [(width, 504)] + [(width, width + 504 if (i - 1) % 4 == 0 and i - 1 > 0 else width) for i in range(1, depth)] + [(1, width)]
        if rgb:
            want += [(256, width), (128, 283), (3, 128)]
        for i, (w, b) in enumerate(zip(ws, bs)):
            if tuple(w.shape) != want[i] or tuple(b.shape) != (want[i][0],):
                raise ValueError("Mip-NeRF 360 MLP layer %d: weight %s / bias %s, expected %s / (%d,)"
                                 % (i, tuple(w.shape), tuple(b.shape), want[i], want[i][0]))
        xf = f32(x0, "x0")
        de = f32(d_enc, "d_enc") if rgb else None
        c = _ctx(xf, lib_ctx)
        wd = [f32(w.detach(), "weight") for w in ws]
        bd = [f32(b.detach(), "bias") for b in bs]
        tab = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        out = torch.empty(rows, 4, device=xf.device)
        tape = torch.empty(c.lib.neo_mip_mlp_train_tape_floats(width, depth, int(rgb), R, n), device=xf.device)
        _lib.check(c.lib.neo_mip_mlp_train_forward(c.handle, width, depth, int(rgb), tab(wd), tab(bd), ptr(xf), ptr(de), R, n, ptr(tape),
                                                   ptr(out), c.stream()))
        ctx_.save_for_backward(xf, tape, out, *wd) if not rgb else ctx_.save_for_backward(xf, tape, out, *wd, de)
        ctx_.meta = (c, width, depth, bool(rgb), R, n, [tuple(w.shape) for w in ws], [tuple(b.shape) for b in bs])
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx_, g):
        c, width, depth, rgb, R, n, wshapes, bshapes = ctx_.meta
        saved = ctx_.saved_tensors
        xf, tape, out = saved[:3]
        wd = list(saved[3:3 + len(wshapes)])
        de = saved[-1] if rgb else None
        dev = xf.device
        g = f32(g.contiguous(), "g_rgbdens")
        gw = [torch.zeros(s, device=dev) for s in wshapes]
        gb = [torch.zeros(s, device=dev) for s in bshapes]
        tab = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        _lib.check(c.lib.neo_mip_mlp_train_backward(c.handle, width, depth, int(rgb), tab(wd), ptr(xf), ptr(de), R, n, ptr(tape), ptr(out),
                                                    ptr(g), tab(gw), tab(gb), c.stream()))
        return (None, None, None, None, None, None, None, *gw, *gb)


def mip_mlp_fused(mlp, x0, d_enc, n, ctx=None):
    """mip_mlp as one native chain each way (round 6): same inputs, same outputs (density (R, n), rgb (R, n, 3)), gradients to
    every parameter tensor.  Shapes outside the chain's range (width not a multiple of 64 or > 1024, depth > 8, a skip after the last
    layer) take the per-layer operators."""
    W, D = mlp.netwidth, len(mlp.pts_linear)
    if W % 64 or W > 1024 or D > 8 or ((D - 1) % 4 == 0 and D - 1 > 0):
        return mip_mlp(mlp, x0, d_enc, n, ctx=ctx)
    layers = mlp.ordered_layers()
    params = [l.weight for l in layers] + [l.bias for l in layers]
    out = _MipTrainMLP.apply(ctx, W, D, not mlp.disable_rgb, n, x0, d_enc, *params)
    R = x0.shape[0] // n
    out = out.reshape(R, n, 4)
    return out[..., 3], out[..., :3]


def mip_render_train(module, batch, train_frac, randomized, near, far, seed=None):
    """models.MipNeRF360.forward WITH autograd / randomized sampling: (renderings, ray_history) as the reference returns them.
    Per level: proposal resampling on the previous level's detached histogram (neo_mip_resample_u: max-dilation, annealed
    softmax cdf, interval sampling; randomized = one jitter per ray, helper.py:358-365, drawn from the library's counter-based
    generator, stream = level), IPE rows (neo_mip_encode), the level's MLP on the linear-layer operators (mip_mlp: gradients
    to every parameter), compositing with a native backward (mip_composite: gradients arrive through the colour AND through
    `weights`, which the interlevel / distortion losses of training_step read).  sdist is detached (stop_level_grad)."""
    from . import ops
    rays_o, rays_d = f32(batch["rays_o"], "rays_o"), f32(batch["rays_d"], "rays_d")
    viewdirs, radii = f32(batch["viewdirs"], "viewdirs"), f32(batch["radii"], "radii")
    dev = rays_o.device
    c = module._context(dev)
    B = rays_o.shape[0]
    counts = (module.num_prop_samples,) * (module.num_levels - 1) + (module.num_nerf_samples,)
    slope = 10.0
    anneal = (slope * float(train_frac)) / ((slope - 1.0) * float(train_frac) + 1.0)
    with torch.no_grad():
        if randomized:
            seed = int(seed) if seed is not None else int(torch.randint(1, 2 ** 62, (1,)).item())
            seed = seed or 1
        sdist = torch.cat([torch.zeros(B, 1, device=dev), torch.ones(B, 1, device=dev)], dim=-1)
        weights = torch.ones(B, 1, device=dev)
        d_enc = ops.pos_enc(viewdirs, 0, 4, ctx=c)                                                # (B,27), append_identity
    prod = 1
    renderings, history = [], []
    for lvl, n in enumerate(counts):
        mlp = module.mlps[lvl]
        dilation = 0.0025 + 0.5 * (1.0 - 0.0) / prod                                             # model.py:266-271
        prod *= n
        with torch.no_grad():
            if randomized:                                                                       # helper.py:358-365
                u_max = _EPS32 + (1 - _EPS32) / n
                max_jitter = (1 - u_max) / (n - 1) - _EPS32
                u = torch.linspace(0, 1 - u_max, n).to(dev)
                jitter = rand_uniform(seed, lvl, B, 1, ctx=c) * max_jitter
            else:                                                                                # helper.py:352-354
                pad = 1 / (2 * n)
                u, jitter = torch.linspace(pad, 1 - pad - _EPS32, n).to(dev), None
            sdist, tdist = mip_resample_u(sdist, weights.detach(), n, near, far, lvl > 0, dilation, anneal, u, jitter, ctx=c)
            x0 = mip_encode(rays_o, rays_d, radii, tdist, mlp.pos_basis_t, ctx=c)
        # the level's MLP as one native chain each way (round 6); `module.train_fused = False`: the per-layer operators (mip_mlp)
        density, rgb = (mip_mlp_fused if getattr(module, "train_fused", True) else mip_mlp)(mlp, x0, d_enc, n, ctx=c)
        weights, colour = mip_composite(rgb, density, tdist, rays_d, 1.0, ctx=c)
        renderings.append({"rgb": colour})
        history.append(dict(density=density, rgb=rgb, sdist=sdist, weights=weights))
    module._raise_flags(c.poll_flags())
    return renderings, history

