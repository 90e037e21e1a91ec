"""The device assertion word without synchronisations (no sync inside the call).

Default `poll_flags = "deferred"`: a chunk loop of forward calls issues ZERO blocking waits, and a ray that misses the
unit sphere (the reference's AssertionError, neo360/helper.py:271) is still raised - by a later call, by
`check_flags()`, or by `render.render_rays_test` before it hands the frame out."""
import pytest
import torch

from neo360_amd import models, render, synth
import cases

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _net():
    net = models.NeRF_TP(num_coarse_samples=16, num_fine_samples=32, num_src_views=cases.NV).to(DEV)
    net.load_state_dict(synth.nerf_tp_state(0))
    sc = cases.small_scene()
    net.set_scene(sc["plane_xz"].to(DEV), sc["plane_xy"].to(DEV), sc["plane_yz"].to(DEV), sc["latent"].to(DEV), sc["image_wh"])
    return net


def _batch(n):
    return {k: v.to(DEV) for k, v in cases.neo_batch(cases.strided_rays(n)).items()}


def _bad(batch, i):
    b = dict(batch)
    b["rays_o"], b["rays_d"] = batch["rays_o"].clone(), batch["rays_d"].clone()
    b["rays_o"][i] = torch.tensor([0.0, 0.0, 5.0], device=DEV)
    b["rays_d"][i] = torch.tensor([1.0, 0.0, 0.0], device=DEV)
    return b


def test_chunk_loop_issues_no_blocking_waits(built_lib):
    net = _net()
    batch = _batch(64)
    net(batch, False, False, 0.0, 0.0, out_depth=True)       # warm-up: uploads, workspaces, constant tables
    net.check_flags()
    ctx = net._context(torch.device(DEV))
    before = ctx.sync_count()
    outs = [net(batch, False, False, 0.0, 0.0, out_depth=True)[1][0] for _ in range(300)]   # the reference's 300 chunk calls
    assert ctx.sync_count() == before, "a forward call of the chunk loop blocked on the device"
    net.check_flags()                                        # one wait, at the end
    assert ctx.sync_count() > before
    assert torch.equal(outs[0], outs[-1])


def test_sphere_miss_is_raised_late_but_raised(built_lib):
    net = _net()
    good, bad = _batch(16), _bad(_batch(16), 5)
    net(good, False, False, 0.0, 0.0, out_depth=True)
    net.check_flags()
    net(bad, False, False, 0.0, 0.0, out_depth=True)         # trips the assertion word on the device; returns at once
    torch.cuda.synchronize()                                 # the posted read has landed in the pinned slot
    with pytest.raises(AssertionError, match="earlier call"):
        net(good, False, False, 0.0, 0.0, out_depth=True)    # the next call looks at completed reads first
    net.check_flags()                                        # nothing left: the word was cleared with the read
    # check_flags() alone also reports it
    net(bad, False, False, 0.0, 0.0, out_depth=True)
    with pytest.raises(AssertionError):
        net.check_flags()
    net(good, False, False, 0.0, 0.0, out_depth=True)
    net.check_flags()


def test_render_rays_test_raises_before_returning(built_lib):
    net = _net()
    with pytest.raises(AssertionError):
        render.render_rays_test(net, _bad(_batch(48), 40), chunk=16)
    out = render.render_rays_test(net, _batch(48), chunk=16)
    assert bool(torch.isfinite(out["rgb"]).all())


def test_more_posts_than_ring_slots(built_lib):
    """64 pinned slots: a caller that never looks still loses nothing (the oldest read is retired into the next take)."""
    net = _net()
    good, bad = _batch(8), _bad(_batch(8), 2)
    net.poll_flags = "never"
    ctx = net._context(torch.device(DEV))
    net(bad, False, False, 0.0, 0.0, out_depth=True)
    ctx.post_flags()
    for _ in range(100):
        ctx.post_flags()
    assert ctx.take_flags(wait=True) & 1


def test_immediate_and_never_modes(built_lib):
    net = _net()
    bad = _bad(_batch(8), 1)
    net.poll_flags = "immediate"
    with pytest.raises(AssertionError):
        net(bad, False, False, 0.0, 0.0, out_depth=True)
    net.poll_flags = "never"
    net(bad, False, False, 0.0, 0.0, out_depth=True)
    net(bad, False, False, 0.0, 0.0, out_depth=True)         # no read in between: no raise
    with pytest.raises(AssertionError):
        net.check_flags()
    net.poll_flags = "bogus"
    with pytest.raises(ValueError):
        net(bad, False, False, 0.0, 0.0, out_depth=True)
