"""GPU: range guard of the split-fp16 ("f16x3") arithmetic.  hi = fp16(x) overflows for |x| >= 65504; the library
must then RAISE (device flag bit 1 -> NeoError), never return plausible numbers; inside the range - including very
large and very small magnitudes - the split kernels must keep fp32-class agreement with the exact fp32-MFMA kernels
and the oracle."""
import pytest
import torch

import cases
import oracle
from conftest import max_abs
from neo360_amd import _lib, models, ops, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _vanilla(state, prec="f16x3"):
    net = models.NeRF().to(DEV)
    net.precision = prec
    net.load_state_dict(state)
    return net


def _rays(n):
    return {k: v.to(DEV) for k, v in cases.strided_rays(n).items()}


def test_vanilla_weight_out_of_range_raises():
    state = synth.vanilla_state(0)
    state["fine_mlp.pts_linears.3.weight"] = state["fine_mlp.pts_linears.3.weight"].clone()
    state["fine_mlp.pts_linears.3.weight"][5, 7] = 7.0e4
    with pytest.raises(_lib.NeoError, match="fp16 range"):
        net = _vanilla(state)
        net(_rays(64), False, False, 0.2, 3.0)
        net.check_flags()
    # the exact fp32 path has no such limit and the flag does not leak into it
    net = _vanilla(state, "f32")
    out = net(_rays(64), False, False, 0.2, 3.0)
    net.check_flags()
    assert bool(torch.isfinite(out[1][0]).all())
    # non-finite weights
    state["fine_mlp.pts_linears.3.weight"][5, 7] = float("nan")
    with pytest.raises(_lib.NeoError):
        net = _vanilla(state)
        net.poll_flags = "immediate"
        net(_rays(64), False, False, 0.2, 3.0)


def test_vanilla_activation_overflow_raises():
    """Weights in range, activations not: every trunk weight x40 grows the activations ~40x per layer."""
    state = {k: (v * 40.0 if "pts_linears" in k and k.endswith("weight") else v) for k, v in synth.vanilla_state(0).items()}
    with pytest.raises(_lib.NeoError, match="fp16 range"):
        net = _vanilla(state)
        net(_rays(64), False, False, 0.2, 3.0)
        net.check_flags()


@pytest.mark.parametrize("scale", [1e3, 1e-4])
def test_neo360_scaled_features_parity_or_flag(scale):
    """Tri-planes and latents scaled x1e3 / x1e-4 (un-normalised encoder outputs): the split evaluators either raise
    the range flag or agree with the oracle to fp32 relative accuracy - on both split kernels."""
    params = synth.nerf_tp_state(0)
    scene = cases.small_scene()
    scene = {k: (v * scale if isinstance(v, torch.Tensor) else v) for k, v in scene.items()}
    cb = cases.neo_batch(cases.strided_rays(96))
    gb = {k: v.to(DEV) for k, v in cb.items()}
    far_c, _ = oracle.rays.sphere_exit_depth(cb["rays_o"], cb["rays_d"])
    tv = torch.linspace(0.05, 0.95, 40)[None, :] * far_c
    rgb, sigma = oracle.neo360.region_eval(params, "fg_fine_mlp.", cb, scene, tv, True, far_c)
    # truth: the oracle in float64.  At x1e3 the fp32 reference arithmetic itself is only good to ~1e-3 relative on the
    # densities (512-term dot products of O(500) features), so the bound is relative to the fp32 oracle's own error.
    dbl = lambda d: {k: (v.double() if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in d.items()}
    rgb64, sigma64 = oracle.neo360.region_eval(dbl(params), "fg_fine_mlp.", dbl(cb), dbl(scene), tv.double(), True, far_c.double())
    rel = lambda a: float(((a.double() - sigma64).abs() / sigma64.abs().clamp_min(1.0)).max())
    cpu_rgb, cpu_sig = float((rgb.double() - rgb64).abs().max()), rel(sigma)
    for preproject in (True, False):
        net = models.NeRF_TP(num_coarse_samples=32, num_fine_samples=64, num_src_views=cases.NV).to(DEV)
        net.load_state_dict(params)
        net.set_scene(scene["plane_xz"].to(DEV), scene["plane_xy"].to(DEV), scene["plane_yz"].to(DEV),
                      scene["latent"].to(DEV), scene["image_wh"], preproject=preproject)
        try:
            got = net.eval_mlp(1, gb, tv.to(DEV), far=far_c.to(DEV)).cpu()
        except _lib.NeoError:
            continue                                                   # flagged: acceptable, never silent
        gpu_rgb, gpu_sig = float((got[..., :3].double() - rgb64).abs().max()), rel(got[..., 3:])
        print("scale %g preproject %s: |rgb - fp64| gpu %.2e cpu32 %.2e; rel sigma gpu %.2e cpu32 %.2e" %
              (scale, preproject, gpu_rgb, cpu_rgb, gpu_sig, cpu_sig))
        assert gpu_rgb < max(1e-4, 2.0 * cpu_rgb)
        assert gpu_sig < 2.0 * cpu_sig + 2e-5, (preproject, gpu_sig, cpu_sig)


def test_neo360_features_beyond_range_raise():
    params = synth.nerf_tp_state(0)
    sc = cases.small_scene()
    gb = {k: v.to(DEV) for k, v in cases.neo_batch(cases.strided_rays(64)).items()}
    far, _ = ops.intersect_sphere(gb["rays_o"], gb["rays_d"])
    tv = torch.linspace(0.05, 0.95, 33, device=DEV)[None, :] * far.reshape(-1, 1)
    for what in ("latent", "plane_xy"):
        for preproject in (True, False):
            scene = dict(sc)
            scene[what] = sc[what] * 1.0e6
            net = models.NeRF_TP(num_coarse_samples=32, num_fine_samples=64, num_src_views=cases.NV).to(DEV)
            net.load_state_dict(params)
            net.set_scene(scene["plane_xz"].to(DEV), scene["plane_xy"].to(DEV), scene["plane_yz"].to(DEV),
                          scene["latent"].to(DEV), scene["image_wh"], preproject=preproject)
            with pytest.raises(_lib.NeoError, match="fp16 range"):
                net.eval_mlp(0, gb, tv, far=far)
            # the word was cleared by the poll: a healthy scene on the same module works afterwards
            net.set_scene(sc["plane_xz"].to(DEV), sc["plane_xy"].to(DEV), sc["plane_yz"].to(DEV), sc["latent"].to(DEV),
                          sc["image_wh"])
            assert bool(torch.isfinite(net.eval_mlp(0, gb, tv, far=far)).all())


def test_c_abi_context_defaults_to_split_arithmetic():
    """include/neo360_hip.h: a fresh neo_ctx computes in the split-fp16 arithmetic (mode 1), like the Python modules."""
    from neo360_amd.context import new_context
    import ctypes
    ctx = new_context(torch.device(DEV))
    net = _vanilla(synth.vanilla_state(0))                  # module default: f16x3
    rays = _rays(32)
    want = net.eval_mlp(1, rays["rays_o"], rays["viewdirs"], torch.linspace(0.3, 2.5, 16, device=DEV)[None].repeat(32, 1))
    layers = net.fine_mlp.ordered_layers()
    ws = [l.weight.detach().contiguous() for l in layers]
    bs = [l.bias.detach().contiguous() for l in layers]
    tab = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    _lib.check(ctx.lib.neo_vanilla_upload_mlp(ctx.handle, 1, tab(ws), tab(bs), ctx.stream()))
    t = torch.linspace(0.3, 2.5, 16, device=DEV)[None].repeat(32, 1).contiguous()
    out = torch.empty(32, 16, 4, device=DEV)
    _lib.check(ctx.lib.neo_vanilla_mlp(ctx.handle, 1, rays["rays_o"].data_ptr(), rays["viewdirs"].data_ptr(), t.data_ptr(),
                                       16, 32, 16, out.data_ptr(), ctx.stream()))
    torch.cuda.synchronize()
    assert torch.equal(out, want)                           # bitwise: the same (split) kernel ran
    ctx.close()
