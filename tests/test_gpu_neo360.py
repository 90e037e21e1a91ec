"""GPU: NeO-360 decoder path (BASELINE config 3) through the drop-in module against
the reference-generated fixtures.  Tolerance 1e-4 abs on rgb / depth."""
import pytest
import torch

import cases
from conftest import max_abs
from neo360_amd import models, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4


def _net(n_coarse, n_fine, gain=1.0):
    net = models.NeRF_TP(num_coarse_samples=n_coarse, num_fine_samples=n_fine, num_src_views=cases.NV).to(DEV)
    net.load_state_dict(synth.nerf_tp_state(0, density_gain=gain))
    sc = cases.small_scene()
    net.set_scene(sc["plane_xz"].to(DEV), sc["plane_xy"].to(DEV), sc["plane_yz"].to(DEV), sc["latent"].to(DEV),
                  sc["image_wh"])
    return net


def _batch(n):
    b = cases.neo_batch(cases.strided_rays(n))
    return {k: v.to(DEV) for k, v in b.items()}


def _render(net, n, chunk):
    batch = _batch(n)
    outs = []
    for i in range(0, n, chunk):
        part = {k: (v if k.startswith("src_") else v[i:i + chunk]) for k, v in batch.items()}
        outs.append(net(part, False, False, 0.0, 0.0, out_depth=True))
    cat = lambda lv, j: torch.cat([o[lv][j] for o in outs]).cpu()
    return dict(rgb0=cat(0, 0), depth0=cat(0, 5), rgb1=cat(1, 0), fg1=cat(1, 1), bg1=cat(1, 2), fgacc1=cat(1, 3),
                lam1=cat(1, 4), depth1=cat(1, 5))


def _check(got, g, depth_tol=TOL):
    for k in ("rgb0", "rgb1", "fg1", "bg1", "fgacc1", "lam1", "depth0", "depth1"):
        assert max_abs(got[k], g[k]) < (depth_tol if k.startswith("depth") else TOL), k


def test_small_two_chunks(golden):
    """300 rays, caller chunk 256: exercises the view-direction tiling quirk and the short last chunk."""
    _check(_render(_net(32, 64), 300, 256), golden("g4_neo_small"))


def test_chunk_dependence_reproduced(golden):
    net = _net(32, 64)
    _check(_render(net, 128, 128), golden("g4_neo_c128"))
    _check(_render(net, 128, 64), golden("g4_neo_c64"))


def test_internal_chunking_equals_callers(golden):
    """One library call with chunk=64 == the caller's own 64-ray loop (whole-frame API)."""
    net = _net(32, 64)
    batch = _batch(128)
    res = net(batch, False, False, 0.0, 0.0, out_depth=True, chunk=64)
    g = golden("g4_neo_c64")
    assert max_abs(res[1][0].cpu(), g["rgb1"]) < TOL and max_abs(res[1][5].cpu(), g["depth1"]) < TOL


def test_sharp_density(golden):
    """Density head x8 (trained-like, peaky weights).  Hierarchical resampling is ill-conditioned
    there: the REFERENCE's own fp32-vs-fp64 noise on depth exceeds 1e-4 with sharp weights
    ("chaotic resampling"), so depth gets the reference's noise floor as tolerance;
    rgb, acc and lambda still meet 1e-4."""
    _check(_render(_net(32, 64, gain=8.0), 256, 256), golden("g4_neo_sharp"), depth_tol=1e-3)


def _check_e2e_default_counts(got, g):
    """End to end at the reference's default 128+256 samples.

    Everything up to and including the fine-level SAMPLE PLACEMENT inputs is strict (1e-4
    on every ray).  The fine level's background branch inverts a cdf whose bins DEscend
    (neo360/model.py:319-331): every new sample is interpolated across the WHOLE [0,1]
    range with t=(u-cdf0)/(cdf1-cdf0), so where a bin carries ~no weight an ulp of the
    fp32 cdf moves a sample anywhere along the ray.  The reference is that sensitive to
    its own rounding (fp32 vs fp64 of the reference differ the same way; its CPU sum
    order even depends on the host's vector width), so bit-level agreement on those rays
    is not defined.  Measured on MI355X vs the fixtures: p99 of |rgb err| 3.6e-6, ~0.2% of
    rays above 1e-4 (max 2.3e-4), all through bg_rgb; fg, acc, lambda, depth stay < 2e-5.
    test_gpu_neo360_stages.py closes the gap: with identical sample positions on both
    sides every ray meets 1e-4."""
    for k in ("rgb0", "depth0", "fg1", "fgacc1", "lam1", "depth1"):
        assert max_abs(got[k], g[k]) < TOL, k
    for k in ("rgb1", "bg1"):
        err = (got[k] - g[k]).abs().amax(dim=-1)
        assert float(err.quantile(0.99)) < 2e-5, k
        assert float((err > TOL).float().mean()) < 0.01, k
        assert float(err.max()) < 2e-3, k
    mse = float(((got["rgb1"].clamp(0, 1) - g["rgb1"].clamp(0, 1)) ** 2).mean())
    assert mse < 1e-10       # PSNR vs the reference frame > 100 dB


def test_reference_sample_counts_1024(golden):
    """One reference-sized chunk: 1024 rays, 128 coarse + 256 fine, fg + bg, 3 views."""
    _check_e2e_default_counts(_render(_net(128, 256), 1024, 1024), golden("g4_neo_1024"))


def test_reference_sample_counts_1500_two_chunks(golden):
    _check_e2e_default_counts(_render(_net(128, 256), 1500, 1024), golden("g4_neo_1500"))


def test_sphere_miss_raises():
    net = _net(32, 64)
    batch = _batch(8)
    batch["rays_o"] = batch["rays_o"].clone()
    batch["rays_o"][3] = torch.tensor([0.0, 0.0, 5.0], device=DEV)
    batch["rays_d"] = batch["rays_d"].clone()
    batch["rays_d"][3] = torch.tensor([1.0, 0.0, 0.0], device=DEV)
    with pytest.raises(AssertionError):
        net(batch, False, False, 0.0, 0.0, out_depth=True)


def test_scene_required():
    net = models.NeRF_TP(num_coarse_samples=32, num_fine_samples=64).to(DEV)
    from neo360_amd import _lib
    with pytest.raises(_lib.NeoError):
        net(_batch(4), False, False, 0.0, 0.0, out_depth=True)
