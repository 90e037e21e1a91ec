"""Bitwise run-to-run repeatability of every point evaluator, and agreement of the split-fp16 kernels
with the exact fp32-MFMA kernels on the same inputs.  A kernel that is fast but returns different
values on each launch is not parity-green: this test found a lanes-48..63 glitch in the split NeO-360
evaluator when it was built with packed-fp32 VALU ops (see neo-360_amd/build.py:EXTRA_FLAGS)."""
import pytest
import torch

import cases
from neo360_amd import models, ops, synth

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda", 0)


def test_vanilla_evaluators_repeatable(built_lib):
    dev = _dev()
    R, N = 4096, 128
    g = torch.Generator().manual_seed(5)
    o = (torch.rand(R, 3, generator=g) * 2 - 1).to(dev)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(dev)
    t = (torch.rand(R, N, generator=g) * 4 + 2).sort(-1).values.to(dev)
    nets = {}
    for prec in ("f32", "f16x3"):
        net = models.NeRF().to(dev)
        net.precision = prec
        net.load_state_dict(synth.vanilla_state(0))
        nets[prec] = net
    with torch.no_grad():
        for level in (0, 1):
            ref = nets["f32"].eval_mlp(level, o, d, t)
            assert torch.equal(ref, nets["f32"].eval_mlp(level, o, d, t))
            runs = [nets["f16x3"].eval_mlp(level, o, d, t) for _ in range(3)]
            assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
            assert (runs[0] - ref).abs().max().item() < 2e-6      # fp32-level agreement, every point


def test_neo360_evaluators_repeatable(built_lib):
    dev = _dev()
    R, NC = 1024, 128
    params = synth.nerf_tp_state(0)
    scene = cases.small_scene()
    gb = {k: v.to(dev) for k, v in cases.neo_batch(cases.strided_rays(R)).items()}

    def mk(prec):
        net = models.NeRF_TP(num_coarse_samples=NC, num_fine_samples=256, num_src_views=cases.NV).to(dev)
        net.precision = prec
        net.load_state_dict(params)
        net.set_scene(scene["plane_xz"].to(dev), scene["plane_xy"].to(dev), scene["plane_yz"].to(dev),
                      scene["latent"].to(dev), scene["image_wh"])
        return net

    with torch.no_grad():
        far, _ = ops.intersect_sphere(gb["rays_o"], gb["rays_d"])
        t_fg = torch.linspace(0.05, 0.95, NC, device=dev)[None, :] * far.reshape(-1, 1)
        t_bg = torch.linspace(0.98, 0.02, NC, device=dev)[None, :].expand(R, NC).contiguous()
        ref_net, h_net, n_net = mk("f32"), mk("f16x3"), mk("f16x3")
        n_net.preproject = False                        # the split evaluator that gathers the 512-channel latent
        for slot, tt in ((0, t_fg), (1, t_fg), (2, t_bg), (3, t_bg)):
            ref = ref_net.eval_mlp(slot, gb, tt, far=far)
            assert torch.equal(ref, ref_net.eval_mlp(slot, gb, tt, far=far))
            for net in (h_net, n_net):
                runs = [net.eval_mlp(slot, gb, tt, far=far) for _ in range(3)]
                assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
                assert (runs[0] - ref).abs().max().item() < 5e-6


def test_mip360_evaluators_repeatable(built_lib):
    dev = _dev()
    R = 512
    batch = {k: v.to(dev) for k, v in cases.mip_rays(R).items()}
    state = synth.mip360_state(0, weight_gain=0.5)
    nets = {}
    for prec in ("f32", "f16x3"):
        net = models.MipNeRF360(num_prop_samples=64, num_nerf_samples=32).to(dev)
        net.precision = prec
        net.load_state_dict(state)
        nets[prec] = net
    with torch.no_grad():
        for slot, n in ((0, 64), (1, 64), (2, 32)):
            s_edges = torch.linspace(0.0, 1.0, n + 1, device=dev)
            tdist = (1.0 / (s_edges / 1e6 + (1.0 - s_edges) / 0.2))[None, :].expand(R, n + 1).contiguous()
            ref = nets["f32"].eval_mlp(slot, batch, tdist)
            assert torch.equal(ref, nets["f32"].eval_mlp(slot, batch, tdist))
            runs = [nets["f16x3"].eval_mlp(slot, batch, tdist) for _ in range(3)]
            assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
            # colours are in [0,1]; densities reach O(10): compare relative to scale
            scale = ref.abs().amax().clamp_min(1.0)
            assert ((runs[0] - ref).abs().max() / scale).item() < 5e-6
