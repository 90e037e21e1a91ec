"""GPU: vanilla NeRF path (BASELINE configs 1-2) through the drop-in module,
against the reference-generated fixtures and the CPU oracle.
Tolerance: 1e-4 abs on rgb / depth (BASELINE.json north_star)."""
import pytest
import torch

import cases
import oracle
from conftest import max_abs
from neo360_amd import models, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4


def _net(gain=1.0):
    net = models.NeRF().to(DEV)
    net.load_state_dict(synth.vanilla_state(0, density_gain=gain))
    return net


def _to(rays):
    return {k: v.to(DEV) for k, v in rays.items()}


def test_mlp_stage_vs_oracle():
    """Fused pos_enc + MLP + activations on arbitrary t (ragged tile: 37*65 points)."""
    net = _net()
    rays = cases.strided_rays(37)
    t = torch.sort(synth.uniform(17, "mlp_t", (37, 65), 0.2, 3.0), dim=-1).values
    got = net.eval_mlp(0, rays["rays_o"].to(DEV), rays["viewdirs"].to(DEV), t.to(DEV)).cpu()
    params = synth.vanilla_state(0)
    pts = oracle.sampling.points_on_rays(t, rays["rays_o"], rays["viewdirs"])
    raw_rgb, raw_sigma = oracle.mlp.vanilla_mlp(params, "coarse_mlp.", oracle.encoding.pos_enc(pts, 0, 10),
                                                oracle.encoding.pos_enc(rays["viewdirs"], 0, 4))
    want = torch.cat([oracle.mlp.colour_activation(raw_rgb), oracle.mlp.density_activation(raw_sigma)], dim=-1)
    assert max_abs(got, want) < 2e-5


@pytest.mark.parametrize("tag,gain", [("", 1.0), ("_sharp", 8.0)])
def test_config1_crop_vs_golden(golden, tag, gain):
    g = golden("g4_vanilla")
    res = _net(gain)(_to(cases.crop_rays(32, 32)), False, False, 0.2, 3.0)
    for lv in (0, 1):
        assert max_abs(res[lv][0].cpu(), g["rgb%d%s" % (lv, tag)]) < TOL
        assert max_abs(res[lv][1].cpu(), g["acc%d%s" % (lv, tag)]) < TOL
        assert max_abs(res[lv][2].cpu(), g["depth%d%s" % (lv, tag)]) < TOL


def test_white_background_ragged(golden):
    g = golden("g4_vanilla")
    res = _net()(_to(cases.strided_rays(200)), False, True, 0.2, 3.0)
    assert max_abs(res[1][0].cpu(), g["rgb1_white"]) < TOL
    assert max_abs(res[1][2].cpu(), g["depth1_white"]) < TOL


def test_chunking_is_invisible():
    """Vanilla results do not depend on the chunk the caller uses (unlike NeO-360)."""
    net = _net()
    rays = _to(cases.strided_rays(300))
    whole = net(rays, False, False, 0.2, 3.0)[1]
    parts = [net({k: v[i:i + 128] for k, v in rays.items()}, False, False, 0.2, 3.0)[1] for i in range(0, 300, 128)]
    assert torch.equal(whole[0], torch.cat([p[0] for p in parts]))
    assert torch.equal(whole[2], torch.cat([p[2] for p in parts]))


def test_single_ray_and_empty():
    net = _net()
    rays = _to(cases.strided_rays(1))
    res = net(rays, False, False, 0.2, 3.0)
    want = oracle.vanilla.render(synth.vanilla_state(0), cases.strided_rays(1), 0.2, 3.0)
    assert max_abs(res[1][0].cpu(), want[1][0]) < TOL
    empty = {k: v[:0] for k, v in rays.items()}
    res = net(empty, False, False, 0.2, 3.0)
    assert res[1][0].shape == (0, 3)


def test_randomized_call_runs():
    """randomized=True is the training call (vanilla_nerf/model.py:281-283): since round 5 it runs on the operators of
    training.py (tests/test_gpu_host_r5.py checks values and gradients); same return structure as the fused call."""
    res = _net()(_to(cases.strided_rays(4)), True, False, 0.2, 3.0, seed=3)
    assert len(res) == 2 and res[1][0].shape == (4, 3) and res[1][1].shape == (4,) and res[1][2].shape == (4,)
    assert bool(torch.isfinite(res[1][0]).all())


def test_weights_reupload_on_change(golden):
    g = golden("g4_vanilla")
    net = _net(8.0)
    rays = _to(cases.crop_rays(32, 32))
    a = net(rays, False, False, 0.2, 3.0)[1][0].cpu()
    net.load_state_dict(synth.vanilla_state(0))
    b = net(rays, False, False, 0.2, 3.0)[1][0].cpu()
    assert max_abs(a, g["rgb1_sharp"]) < TOL and max_abs(b, g["rgb1"]) < TOL


def test_full_frame_properties():
    """BASELINE config 2 size (640x480, 64+128): size-independent properties —
    finite outputs, acc in [0,1], depth within [near, far], and a seeded strip
    that agrees with the oracle."""
    from neo360_amd import ops
    net = _net()
    H, W = 480, 640
    c2w = synth.look_at_origin(40.0)
    ro, vd, rd, _ = ops.get_ray_directions_and_rays(H, W, 0.8 * W, c2w)
    res = net(dict(rays_o=ro, viewdirs=vd, rays_d=rd), False, False, 0.2, 3.0)
    rgb, acc, depth = res[1]
    assert rgb.shape == (H * W, 3) and bool(torch.isfinite(rgb).all()) and bool(torch.isfinite(depth).all())
    assert float(acc.min()) >= 0.0 and float(acc.max()) <= 1.0 + 1e-5
    assert float(depth.min()) >= 0.0 and float(depth.max()) <= 3.0 + 1e-4
    rows = torch.arange(200 * W, 200 * W + 96, device=DEV)
    strip = {k: v[rows].cpu() for k, v in dict(rays_o=ro, viewdirs=vd, rays_d=rd).items()}
    want = oracle.vanilla.render(synth.vanilla_state(0), strip, 0.2, 3.0)
    assert max_abs(rgb[rows].cpu(), want[1][0]) < TOL
    assert max_abs(depth[rows].cpu(), want[1][2]) < TOL


@pytest.mark.parametrize("gain", [1.0, 8.0])
def test_f16x3_split_path(golden, gain):
    """The fp16-matrix-core path (hi/lo-split operands) against the fixtures, the oracle MLP stage and
    the exact-fp32 MFMA path: same 1e-4 contract, fp32-class error."""
    g = golden("g4_vanilla")
    tag = "" if gain == 1.0 else "_sharp"
    net = _net(gain)
    net.precision = "f16x3"
    rays = _to(cases.crop_rays(32, 32))
    res = net(rays, False, False, 0.2, 3.0)
    for lv in (0, 1):
        assert max_abs(res[lv][0].cpu(), g["rgb%d%s" % (lv, tag)]) < TOL
        assert max_abs(res[lv][2].cpu(), g["depth%d%s" % (lv, tag)]) < TOL
    # stage level, vs the oracle and vs the fp32-MFMA kernel on identical points
    r = cases.strided_rays(37)
    t = torch.sort(synth.uniform(17, "mlp_t", (37, 65), 0.2, 3.0), dim=-1).values
    got = net.eval_mlp(0, r["rays_o"].to(DEV), r["viewdirs"].to(DEV), t.to(DEV)).cpu()
    ref32 = _net(gain)
    ref32.precision = "f32"
    exact = ref32.eval_mlp(0, r["rays_o"].to(DEV), r["viewdirs"].to(DEV), t.to(DEV)).cpu()
    assert max_abs(got, exact) < 1e-5
