"""GPU: PixelNeRF baseline decoder (vanilla_nerf/model_pixel.py:133-258) — HIP path vs the oracle and vs the
reference-generated fixture g7 (stage-level and end to end, incl. the chunk-dependent direction tiling)."""
import pytest
import torch

import cases
import oracle
from conftest import max_abs
from neo360_amd import models, render, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4
PER_RAY = ("rays_o", "rays_d", "viewdirs")


def _net(gain=1.0, preproject=True):
    scene = cases.small_scene()
    net = models.PixelNeRF(num_src_views=cases.NV).to(DEV)
    net.preproject = preproject                 # True: latent pre-projected through the first layer (default); False: the reference's order
    net.load_state_dict(synth.pixelnerf_state(0, density_gain=gain))
    net.set_scene(scene["latent"].to(DEV), scene["image_wh"])
    return net, scene


@pytest.mark.parametrize("preproject", [True, False])
def test_mlp_stage_matches_oracle_and_reproduces_direction_tiling(preproject):
    net, scene = _net(preproject=preproject)
    params = synth.pixelnerf_state(0)
    batch = cases.neo_batch(cases.strided_rays(128))
    gb = {k: v.to(DEV) for k, v in batch.items()}
    t = torch.sort(torch.rand(128, 97, generator=torch.Generator().manual_seed(3)) * 2.3 + 0.2, dim=-1).values
    for slot, prefix in ((0, "coarse_mlp."), (1, "fine_mlp.")):
        got = net.eval_mlp(slot, gb, t.to(DEV)).cpu()
        rgb, sigma = oracle.pixelnerf.region_eval(params, prefix, batch, scene, t)
        assert max_abs(got[..., :3], rgb) < 2e-5 and max_abs(got[..., 3:], sigma) < 2e-5, prefix
    # two chunk sizes give different colours (reference quirk), identical densities; each matches the oracle
    a = net.eval_mlp(0, gb, t.to(DEV), chunk=128).cpu()
    b = net.eval_mlp(0, gb, t.to(DEV), chunk=64).cpu()
    assert max_abs(a[..., :3], b[..., :3]) > 1e-4 and max_abs(a[..., 3], b[..., 3]) == 0.0
    halves = [oracle.pixelnerf.region_eval(params, "coarse_mlp.", {k: (v[i:i + 64] if k in PER_RAY else v) for k, v in batch.items()},
                                           scene, t[i:i + 64])[0] for i in (0, 64)]
    assert max_abs(b[..., :3], torch.cat(halves)) < 2e-5
    # bitwise repeatable
    assert torch.equal(net.eval_mlp(0, gb, t.to(DEV)).cpu(), net.eval_mlp(0, gb, t.to(DEV)).cpu())


@pytest.mark.parametrize("tag,n_rays,chunk,gain,white", [("a", 300, 256, 1.0, False), ("sharp", 128, 128, 8.0, False),
                                                         ("white", 96, 96, 1.0, True)])
@pytest.mark.parametrize("preproject", [True, False])
def test_end_to_end_vs_reference_fixture(golden, tag, n_rays, chunk, gain, white, preproject):
    g = golden("g7_pixelnerf")
    net, _ = _net(gain, preproject)
    batch = {k: v.to(DEV) for k, v in cases.neo_batch(cases.strided_rays(n_rays)).items()}
    got = {k: [] for k in ("rgb0", "acc0", "depth0", "rgb1", "acc1", "depth1")}
    for i in range(0, n_rays, chunk):
        part = {k: (v[i:i + chunk] if k in PER_RAY else v) for k, v in batch.items()}
        res = net(part, False, white, 0.2, 2.5)
        for lv in (0, 1):
            got["rgb%d" % lv].append(res[lv][0]); got["acc%d" % lv].append(res[lv][1]); got["depth%d" % lv].append(res[lv][2])
    for k, v in got.items():
        assert max_abs(torch.cat(v).cpu(), g["%s_%s" % (k, tag)]) < TOL, (k, tag)
    # the whole-frame call with the chunk passed down reproduces the chunk loop bit for bit
    whole = render.render_rays_test(net, batch, chunk=chunk, white_bkgd=white, near=0.2, far=2.5)
    assert max_abs(whole["rgb"], torch.cat(got["rgb1"])) == 0.0 and max_abs(whole["depth"], torch.cat(got["depth1"])) == 0.0


def test_preprojection_is_a_reassociation():
    """Both operation orders agree to fp32 rounding on the per-point outputs, and new weights / a new latent rebuild the
    projected map (the module re-uploads on parameter changes; the library tracks scene and weight epochs)."""
    a, scene = _net(preproject=True)
    b, _ = _net(preproject=False)
    gb = {k: v.to(DEV) for k, v in cases.neo_batch(cases.strided_rays(64)).items()}
    t = torch.sort(torch.rand(64, 65, generator=torch.Generator().manual_seed(4)) * 2.3 + 0.2, dim=-1).values.to(DEV)
    ya, yb = a.eval_mlp(1, gb, t), b.eval_mlp(1, gb, t)
    assert 0.0 < max_abs(ya, yb) < 5e-6
    with torch.no_grad():
        for net in (a, b):
            net.fine_mlp.pts_linears[0].weight.mul_(1.25)
    ya2, yb2 = a.eval_mlp(1, gb, t), b.eval_mlp(1, gb, t)
    assert max_abs(ya2, ya) > 1e-4 and max_abs(ya2, yb2) < 5e-6
    for net in (a, b):
        net.set_scene(scene["latent"].to(DEV) * 0.5, scene["image_wh"])
    ya3, yb3 = a.eval_mlp(1, gb, t), b.eval_mlp(1, gb, t)
    assert max_abs(ya3, ya2) > 1e-4 and max_abs(ya3, yb3) < 5e-6
