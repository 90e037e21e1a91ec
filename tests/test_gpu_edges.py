"""GPU: edge cases of the render entry points — empty batches, a single ray, and the largest supported
sample counts (n_coarse + 1 + n_fine = 1024: the 1024-wide in-LDS sort of the resampler)."""
import pytest
import torch

import cases
import oracle
from conftest import max_abs
from neo360_amd import models, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
PER_RAY = ("rays_o", "rays_d", "viewdirs")


def _take(batch, n):
    return {k: (v[:n] if k in PER_RAY else v) for k, v in batch.items()}


def test_neo360_single_ray_and_empty():
    scene = cases.small_scene()
    net = models.NeRF_TP(num_coarse_samples=32, num_fine_samples=64, num_src_views=cases.NV).to(DEV)
    params = synth.nerf_tp_state(0)
    net.load_state_dict(params)
    net.set_scene(scene["plane_xz"].to(DEV), scene["plane_xy"].to(DEV), scene["plane_yz"].to(DEV),
                  scene["latent"].to(DEV), scene["image_wh"])
    batch = cases.neo_batch(cases.strided_rays(4))
    one = _take(batch, 1)
    got = net({k: v.to(DEV) for k, v in one.items()}, False, False, 0.0, 0.0, out_depth=True)
    want = oracle.neo360.render(params, one, scene, 32, 64)
    assert max_abs(got[1][0], want[1][0]) < 1e-4 and max_abs(got[1][5], want[1][5]) < 1e-4
    none = net({k: v.to(DEV) for k, v in _take(batch, 0).items()}, False, False, 0.0, 0.0, out_depth=True)
    assert none[1][0].shape == (0, 3) and none[1][5].shape == (0,)


def test_pixelnerf_single_ray_and_empty():
    scene = cases.small_scene()
    net = models.PixelNeRF(num_src_views=cases.NV).to(DEV)
    params = synth.pixelnerf_state(0)
    net.load_state_dict(params)
    net.set_scene(scene["latent"].to(DEV), scene["image_wh"])
    batch = cases.neo_batch(cases.strided_rays(4))
    one = _take(batch, 1)
    got = net({k: v.to(DEV) for k, v in one.items()}, False, False, 0.2, 2.5)
    want = oracle.pixelnerf.render(params, one, scene, 0.2, 2.5)
    assert max_abs(got[1][0], want[1][0]) < 1e-4 and max_abs(got[1][2], want[1][2]) < 1e-4
    none = net({k: v.to(DEV) for k, v in _take(batch, 0).items()}, False, False, 0.2, 2.5)
    assert none[1][0].shape == (0, 3) and none[1][2].shape == (0,)


def test_vanilla_largest_sample_counts():
    """256 coarse + 767 fine = 1024 fine-level samples per ray."""
    net = models.NeRF(num_coarse_samples=256, num_fine_samples=767).to(DEV)
    params = synth.vanilla_state(0)
    net.load_state_dict(params)
    rays = cases.strided_rays(6)
    got = net({k: v.to(DEV) for k, v in rays.items()}, False, False, 0.2, 3.0)
    want = oracle.vanilla.render(params, rays, 0.2, 3.0, 256, 767)
    assert max_abs(got[0][0], want[0][0]) < 1e-5
    assert max_abs(got[1][0], want[1][0]) < 1e-4 and max_abs(got[1][2], want[1][2]) < 2e-4
    with pytest.raises(Exception):
        models.NeRF(num_coarse_samples=256, num_fine_samples=768).to(DEV)({k: v.to(DEV) for k, v in rays.items()},
                                                                         False, False, 0.2, 3.0)


def test_pos_enc_large_and_non_finite_arguments():
    """The encodings use a Cody-Waite sine for |a| <= 65536 and an fp64 range reduction up to |a| < 2^40
    (csrc/common.h:sin_cw), where a = 2^k x is the fp32 argument the reference hands to torch.sin (helper.py:121-125).
    Beyond 2^40 the fp32 spacing of a exceeds the period 60,000-fold; the kernels return 0 for finite |a| >= 4e18 and
    NaN for inf / NaN (documented deviation: torch.sin still returns the sine of the rounded argument there)."""
    from neo360_amd import ops
    g = torch.Generator().manual_seed(5)
    mags = torch.tensor([1e-3, 1.0, 100.0, 6.5e4, 7e4, 1e6, 3e7, 2e9])           # x 512 < 2^40
    x = ((torch.rand(64, mags.numel(), 3, generator=g) * 2 - 1) * mags[None, :, None]).reshape(-1, 3)
    got = ops.pos_enc(x.to(DEV), 0, 10).cpu()
    # reference arithmetic: fp32 scaled argument (exact: power-of-two scales), fp32 phase add, sine of THAT argument
    exact = torch.cat([x.double()] + [torch.sin((x * 2.0 ** k).double()) for k in range(10)]
                      + [torch.sin((x * 2.0 ** k + 0.5 * torch.pi).double()) for k in range(10)], dim=-1)
    assert got.shape == exact.shape
    assert torch.equal(got[:, :3], x)
    assert max_abs(got[:, 3:], exact[:, 3:].float()) < 2e-7
    want_torch = oracle.encoding.pos_enc(x, 0, 10)                                  # the CPU library sine agrees as well
    assert max_abs(got, want_torch) < 2e-7
    huge = torch.tensor([[1e19, -3e30, 1e37]])
    out = ops.pos_enc(huge.to(DEV), 0, 4).cpu()
    assert torch.isfinite(out).all() and float(out[:, 3:].abs().max()) == 0.0
    bad = torch.tensor([[float("inf"), float("nan"), -float("inf")]])
    out = ops.pos_enc(bad.to(DEV), 0, 4).cpu()
    assert torch.isnan(out[:, 3:]).all()
