"""GPU: host-side behaviours added in round 4.

* `render_rays_test(on_range="retry_f32")`: a frame that trips the range guard of the split-fp16 arithmetic is
  re-rendered on the exact fp32 kernels instead of failing (the reference is plain fp32 and never fails at any
  magnitude, neo360/model.py:343-407) - VERDICT r3 task 5.
* One context driven from two streams: launches are ordered on the device (tp_dirsum / workspaces are
  per-context scratch).
* `training.gather_features` re-uploads when the maps are NEW tensors at a recycled address.
* `close()` turns unread deferred assertions into a RuntimeWarning.
"""
import warnings

import pytest
import torch

import cases
from neo360_amd import _lib, models, ops, render, synth, training

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _net(scene, **kw):
    net = models.NeRF_TP(num_coarse_samples=16, num_fine_samples=32, num_src_views=cases.NV, **kw).to(DEV)
    net.load_state_dict(synth.nerf_tp_state(0))
    net.set_scene(scene["plane_xz"].to(DEV), scene["plane_xy"].to(DEV), scene["plane_yz"].to(DEV), scene["latent"].to(DEV),
                  scene["image_wh"])
    return net


def _batch(n):
    return {k: v.to(DEV) for k, v in cases.neo_batch(cases.strided_rays(n)).items()}


@pytest.mark.parametrize("what", ["latent", "plane_xy"])
def test_range_guard_hit_is_rerendered_exactly(what):
    """x1e6 features leave the fp16 range: the default frame call returns the exact kernels' frame, bitwise, and says so."""
    sc = dict(cases.small_scene())
    sc[what] = sc[what] * 1.0e6
    batch = _batch(200)
    exact = _net(sc)
    exact.precision = "f32"
    want = render.render_rays_test(exact, batch, chunk=64)
    assert "precision_used" not in want

    net = _net(sc)
    with pytest.warns(RuntimeWarning, match="re-rendered on the exact fp32 kernels"):
        got = render.render_rays_test(net, batch, chunk=64)
    assert got["precision_used"] == "f32"
    for k in ("rgb", "depth", "acc", "fg_rgb", "bg_rgb"):
        assert torch.equal(got[k], want[k]), k
    assert net.precision is None                     # the module's own setting is untouched
    # the downgrade is per frame, the warning once per module: the next frame trips the guard again and is exact again
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        again = render.render_rays_test(net, batch, chunk=64)
    assert again["precision_used"] == "f32" and torch.equal(again["rgb"], want["rgb"])
    # a healthy scene on the same module goes back to the split kernels
    ok = cases.small_scene()
    net.set_scene(ok["plane_xz"].to(DEV), ok["plane_xy"].to(DEV), ok["plane_yz"].to(DEV), ok["latent"].to(DEV), ok["image_wh"])
    fine = render.render_rays_test(net, batch, chunk=64)
    assert "precision_used" not in fine and bool(torch.isfinite(fine["rgb"]).all())
    # on_range="raise" keeps the fail-stop behaviour
    bad = _net(sc)
    with pytest.raises(_lib.NeoRangeError, match="fp16 range"):
        render.render_rays_test(bad, batch, chunk=64, on_range="raise")


def test_vanilla_range_guard_retry():
    state = {k: (v * 40.0 if "pts_linears" in k and k.endswith("weight") else v) for k, v in synth.vanilla_state(0).items()}
    rays = {k: v.to(DEV) for k, v in cases.strided_rays(96).items()}
    exact = models.NeRF().to(DEV)
    exact.precision = "f32"
    exact.load_state_dict(state)
    want = render.render_rays_test(exact, rays)
    net = models.NeRF().to(DEV)
    net.load_state_dict(state)
    with pytest.warns(RuntimeWarning):
        got = render.render_rays_test(net, rays)
    assert got["precision_used"] == "f32" and torch.equal(got["rgb"], want["rgb"]) and torch.equal(got["depth"], want["depth"])


def test_two_streams_one_context_are_ordered():
    """Two evaluator launches of ONE module on two streams share the per-launch direction table and the projected maps:
    the context orders them on the device, results equal the single-stream ones."""
    sc = cases.small_scene()
    net = _net(sc)
    b1, b2 = _batch(512), _batch(384)
    b2["viewdirs"] = torch.nn.functional.normalize(b2["viewdirs"] + 0.3, dim=-1)
    far1, _ = ops.intersect_sphere(b1["rays_o"], b1["rays_d"])
    far2, _ = ops.intersect_sphere(b2["rays_o"], b2["rays_d"])
    t1 = torch.linspace(0.05, 0.95, 97, device=DEV)[None, :] * far1.reshape(-1, 1)
    t2 = torch.linspace(0.05, 0.95, 65, device=DEV)[None, :] * far2.reshape(-1, 1)
    want1, want2 = net.eval_mlp(1, b1, t1, far=far1), net.eval_mlp(1, b2, t2, far=far2)
    ctx = net._context(torch.device(DEV))
    before = ctx.stream_waits()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for _ in range(8):
        with torch.cuda.stream(s1):
            a = net.eval_mlp(1, b1, t1, far=far1)
        with torch.cuda.stream(s2):
            b = net.eval_mlp(1, b2, t2, far=far2)
        outs.append((a, b))
    torch.cuda.synchronize()
    assert ctx.stream_waits() > before
    for a, b in outs:
        assert torch.equal(a, want1) and torch.equal(b, want2)


def test_gather_features_reuploads_fresh_maps_at_recycled_addresses():
    """A training loop's encoder emits NEW map tensors every step, at version 0 and very likely at the addresses the
    previous step's (freed) maps had: the device copy must follow the tensors, not the addresses."""
    sc = cases.small_scene()
    net = _net(sc)
    batch = _batch(32)
    pts = (torch.rand(256, 3, device=DEV) - 0.5)
    seen = []
    for step in range(3):
        maps = [sc[k].to(DEV) * (1.0 + step) for k in ("plane_xz", "plane_xy", "plane_yz", "latent")]
        ptrs = tuple(m.data_ptr() for m in maps)
        with torch.enable_grad():
            world, local = training.gather_features(net, pts, *maps, batch)
        seen.append((ptrs, world.clone(), local.clone()))
        del maps, world, local                                       # freed: the allocator may hand the blocks out again
    recycled = seen[1][0] == seen[0][0] or seen[2][0] == seen[1][0]
    for step in (1, 2):
        scale = (1.0 + step) / 1.0
        assert torch.allclose(seen[step][1], seen[0][1] * scale, rtol=1e-5, atol=1e-6), "stale device copy of the planes"
        assert torch.allclose(seen[step][2], seen[0][2] * scale, rtol=1e-5, atol=1e-6), "stale device copy of the latent"
    print("addresses recycled between steps:", recycled)
    # the same tensor objects, unchanged: no re-upload (identity + version match)
    maps = [sc[k].to(DEV) for k in ("plane_xz", "plane_xy", "plane_yz", "latent")]
    with torch.enable_grad():
        training.gather_features(net, pts, *maps, batch)
    assert net.scene_matches(maps)
    maps[3].mul_(2.0)                                                 # in-place edit: version bump
    assert not net.scene_matches(maps)


def test_set_scene_failure_leaves_no_matching_fingerprint():
    sc = cases.small_scene()
    net = _net(sc)
    maps = [sc[k].to(DEV) for k in ("plane_xz", "plane_xy", "plane_yz")] + [sc["latent"]]     # latent on the CPU: rejected
    with pytest.raises(_lib.NeoError):
        net.set_scene(*maps, sc["image_wh"])
    assert not net.scene_matches(maps)
    with pytest.raises(_lib.NeoError, match="no scene features"):
        net(_batch(8), False, False, 0.0, 0.0, out_depth=True)


def test_close_warns_about_unread_assertions():
    net = _net(cases.small_scene())
    b = _batch(16)
    b["rays_o"] = b["rays_o"].clone()
    b["rays_d"] = b["rays_d"].clone()
    b["rays_o"][3] = torch.tensor([0.0, 0.0, 5.0], device=DEV)      # misses the unit sphere
    b["rays_d"][3] = torch.tensor([1.0, 0.0, 0.0], device=DEV)
    net(b, False, False, 0.0, 0.0, out_depth=True)                   # deferred: returns at once, nobody checks
    with pytest.warns(RuntimeWarning, match="unread device assertions"):
        net.close()
