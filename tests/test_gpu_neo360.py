"""GPU: NeO-360 decoder path (BASELINE config 3) through the drop-in module against
the reference-generated fixtures.  Tolerance 1e-4 abs on rgb / depth."""
import pytest
import torch

import cases
from conftest import check_vs_reference_noise, max_abs, record_parity
from neo360_amd import models, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4


def _net(n_coarse, n_fine, gain=1.0, nv=cases.NV, preproject=True):
    net = models.NeRF_TP(num_coarse_samples=n_coarse, num_fine_samples=n_fine, num_src_views=nv).to(DEV)
    net.load_state_dict(synth.nerf_tp_state(0, density_gain=gain))
    sc = cases.small_scene(nv=nv)
    net.set_scene(sc["plane_xz"].to(DEV), sc["plane_xy"].to(DEV), sc["plane_yz"].to(DEV), sc["latent"].to(DEV),
                  sc["image_wh"], preproject=preproject)
    return net


def _batch(n, nv=cases.NV):
    b = cases.neo_batch(cases.strided_rays(n), nv=nv)
    return {k: v.to(DEV) for k, v in b.items()}


def _render(net, n, chunk, nv=cases.NV, white=False):
    batch = _batch(n, nv)
    outs = []
    for i in range(0, n, chunk):
        part = {k: (v if k.startswith("src_") else v[i:i + chunk]) for k, v in batch.items()}
        outs.append(net(part, False, white, 0.0, 0.0, out_depth=True))
    cat = lambda lv, j: torch.cat([o[lv][j] for o in outs]).cpu()
    return dict(rgb0=cat(0, 0), depth0=cat(0, 5), rgb1=cat(1, 0), fg1=cat(1, 1), bg1=cat(1, 2), fgacc1=cat(1, 3),
                lam1=cat(1, 4), depth1=cat(1, 5))


def _check(got, g, depth_tol=TOL, label=None):
    errs = {k: max_abs(got[k], g[k]) for k in ("rgb0", "rgb1", "fg1", "bg1", "fgacc1", "lam1", "depth0", "depth1")}
    if label:
        record_parity("neo360_e2e/" + label, **{"max_" + k: v for k, v in errs.items()})
    for k, e in errs.items():
        assert e < (depth_tol if k.startswith("depth") else TOL), (k, e)


def test_small_two_chunks(golden):
    """300 rays, caller chunk 256: exercises the view-direction tiling quirk and the short last chunk."""
    _check(_render(_net(32, 64), 300, 256), golden("g4_neo_small"), label="small_two_chunks")


def test_chunk_dependence_reproduced(golden):
    net = _net(32, 64)
    _check(_render(net, 128, 128), golden("g4_neo_c128"), label="chunk128")
    _check(_render(net, 128, 64), golden("g4_neo_c64"), label="chunk64")


def test_internal_chunking_equals_callers(golden):
    """One library call with chunk=64 == the caller's own 64-ray loop (whole-frame API)."""
    net = _net(32, 64)
    batch = _batch(128)
    res = net(batch, False, False, 0.0, 0.0, out_depth=True, chunk=64)
    g = golden("g4_neo_c64")
    assert max_abs(res[1][0].cpu(), g["rgb1"]) < TOL and max_abs(res[1][5].cpu(), g["depth1"]) < TOL


def _check_vs_reference_noise(got, g, noise, label, flip=None):
    check_vs_reference_noise(got, g, noise, "neo360_e2e/" + label, flip=flip)


def test_sharp_density(golden, golden_optional):
    """Density head x8 (trained-like, peaky weights): hierarchical resampling is ill-conditioned there in the
    reference itself (fixture g4_neo_sharp_noise: depth1 differs by up to 4.4e-4 between its fp32 and fp64 runs)."""
    _check_vs_reference_noise(_render(_net(32, 64, gain=8.0), 256, 256), golden("g4_neo_sharp"),
                              golden("g4_neo_sharp_noise"), "sharp", flip=golden_optional("g4_neo_sharp_flip"))


@pytest.mark.parametrize("preproject", [3, True, 2, False])
def test_reference_sample_counts_1024(golden, golden_optional, preproject):
    """One reference-sized chunk: 1024 rays, 128 coarse + 256 fine, fg + bg, 3 views; both split evaluators
    (latent pre-projected through the first-layer weights = default, and the reference's operation order)."""
    _check_vs_reference_noise(_render(_net(128, 256, preproject=preproject), 1024, 1024), golden("g4_neo_1024"),
                              golden("g4_neo_1024_noise"), "1024 preproject=%s" % preproject,
                              flip=golden_optional("g4_neo_1024_flip"))


def test_reference_sample_counts_1500_two_chunks(golden, golden_optional):
    _check_vs_reference_noise(_render(_net(128, 256), 1500, 1024), golden("g4_neo_1500"),
                              golden("g4_neo_1500_noise"), "1500", flip=golden_optional("g4_neo_1500_flip"))


def test_preprojection_is_a_reassociation(golden):
    """Per-point outputs of the pre-projected evaluator vs the evaluator that gathers the 512-channel latent and
    multiplies it per point: identical up to fp32 reassociation (W.bilerp(F) = bilerp(W.F)), on every slot."""
    from neo360_amd import ops
    a, b, c = _net(32, 64, preproject=True), _net(32, 64, preproject=False), _net(32, 64, preproject=2)
    batch = _batch(256)
    far, _ = ops.intersect_sphere(batch["rays_o"], batch["rays_d"])
    t_fg = torch.linspace(0.03, 0.97, 65, device=DEV)[None, :] * far.reshape(-1, 1)
    t_bg = torch.linspace(0.99, 0.01, 65, device=DEV)[None, :].expand(256, 65).contiguous()
    for slot, tv in ((0, t_fg), (1, t_fg), (2, t_bg), (3, t_bg)):
        ya, yb = a.eval_mlp(slot, batch, tv, far=far), b.eval_mlp(slot, batch, tv, far=far)
        assert max_abs(ya[..., :3], yb[..., :3]) < 5e-6, slot
        assert max_abs(ya[..., 3], yb[..., 3]) < 2e-5, slot          # softplus densities reach O(10)
        # the same reassociation applied to the three tri-planes (world columns): mlp_tp_hpp.hip
        yc = c.eval_mlp(slot, batch, tv, far=far)
        assert max_abs(yc[..., :3], yb[..., :3]) < 5e-6, slot
        assert max_abs(yc[..., 3], yb[..., 3]) < 2e-5, slot
    # new weights invalidate the projection
    sd = synth.nerf_tp_state(3)
    a.load_state_dict(sd); b.load_state_dict(sd); c.load_state_dict(sd)
    ya, yb = a.eval_mlp(1, batch, t_fg, far=far), b.eval_mlp(1, batch, t_fg, far=far)
    assert max_abs(ya, yb) < 2e-5
    assert max_abs(c.eval_mlp(1, batch, t_fg, far=far), yb) < 2e-5
    # and so does a new scene
    sc = cases.small_scene(seed=11)
    for net in (a, b, c):
        net.set_scene(sc["plane_xz"].to(DEV), sc["plane_xy"].to(DEV), sc["plane_yz"].to(DEV), sc["latent"].to(DEV), sc["image_wh"])
    ya, yb = a.eval_mlp(3, batch, t_bg, far=far), b.eval_mlp(3, batch, t_bg, far=far)
    assert max_abs(ya, yb) < 2e-5
    assert c.preproject == 2 and max_abs(c.eval_mlp(3, batch, t_bg, far=far), yb) < 2e-5


@pytest.mark.parametrize("nv", [1, 2, 5])
def test_other_view_counts(golden, nv):
    """NeRF_TP(num_src_views=1|2|5) against fixtures generated by the reference with that many source views."""
    got = _render(_net(32, 64, nv=nv), 96, 96, nv=nv)
    _check(got, golden("g4_neo_nv%d" % nv))


def test_white_bkgd_is_ignored_with_out_depth(golden):
    """The evaluation call (out_depth=True) composites with white_bkgd=False whatever the caller passes
    (neo360/model.py:487-517): results must not change."""
    net = _net(32, 64)
    a, b = _render(net, 300, 256, white=False), _render(net, 300, 256, white=True)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    _check(b, golden("g4_neo_small"))


def test_out_of_range_lookups():
    """Sample positions far outside the feature volumes / image frusta: every tap that falls off a map contributes
    zero (grid_sample padding_mode='zeros'), per-point outputs still match the oracle."""
    import oracle
    net = _net(32, 64)
    params, scene = synth.nerf_tp_state(0), cases.small_scene()
    cb = cases.neo_batch(cases.strided_rays(64))
    gb = {k: v.to(DEV) for k, v in cb.items()}
    far_c, _ = oracle.rays.sphere_exit_depth(cb["rays_o"], cb["rays_d"])
    tv = torch.linspace(0.0, 6.0, 48)[None, :].repeat(64, 1)           # up to 6 units along the ray: far off every map
    got = net.eval_mlp(0, gb, tv.to(DEV), far=far_c.to(DEV)).cpu()
    rgb, sigma = oracle.neo360.region_eval(params, "fg_coarse_mlp.", cb, scene, tv, True, far_c)
    assert max_abs(got[..., :3], rgb) < 2e-5 and max_abs(got[..., 3:], sigma) < 2e-5
    # behind / beside the source cameras as well: rays shot away from the scene
    cb2 = dict(cb)
    cb2["rays_d"] = -cb["rays_d"]
    cb2["viewdirs"] = -cb["viewdirs"]
    gb2 = {k: v.to(DEV) for k, v in cb2.items()}
    got = net.eval_mlp(1, gb2, tv.to(DEV), far=far_c.to(DEV)).cpu()
    rgb, sigma = oracle.neo360.region_eval(params, "fg_fine_mlp.", cb2, scene, tv, True, far_c)
    assert max_abs(got[..., :3], rgb) < 2e-5 and max_abs(got[..., 3:], sigma) < 2e-5


class _StubEncoder(torch.nn.Module):
    """Stands in for GridEncoder: deterministic features that depend on the source images, so a stale cache shows."""

    class _Spatial:
        latent = None

    def __init__(self):
        super().__init__()
        self.gain = torch.nn.Parameter(torch.ones(()))
        self.spatial_encoder = self._Spatial()
        self.calls = 0

    def forward(self, src_imgs, src_poses, src_focal, src_c):
        self.calls += 1
        sc = cases.small_scene()
        k = float(src_imgs.mean()) + 1.0
        self.spatial_encoder.latent = (sc["latent"].to(src_imgs.device) * k * self.gain).contiguous()
        return tuple((sc[n].to(src_imgs.device) * k * self.gain).contiguous() for n in ("plane_xz", "plane_xy", "plane_yz"))


def test_attached_encoder_runs_once_per_scene():
    """INTEGRATION.md's 'no edits' mode: an attached encoder is run once per distinct src_imgs, re-run for a new
    tensor (even one the allocator places at the old address), an in-place edit, or new encoder weights."""
    enc = _StubEncoder().to(DEV)
    net = models.NeRF_TP(num_coarse_samples=32, num_fine_samples=64, num_src_views=cases.NV, encoder=enc).to(DEV)
    net.load_state_dict(synth.nerf_tp_state(0), strict=False)
    batch = _batch(64)
    chunks = [{k: (v if k.startswith("src_") else v[i:i + 32]) for k, v in batch.items()} for i in (0, 32)]
    a = [net(c, False, False, 0.0, 0.0, out_depth=True)[1][0] for c in chunks]
    assert enc.calls == 1                                        # two chunks of one frame: encoded once
    ref = _net(32, 64)(batch, False, False, 0.0, 0.0, out_depth=True, chunk=32)[1][0]
    assert max_abs(torch.cat(a), ref) < 1e-6                     # src_imgs = 0 -> k = 1: the plain small scene
    # a new scene in a NEW tensor at (very likely) the same address
    shape = batch["src_imgs"].shape
    del batch["src_imgs"]
    for c in chunks:
        del c["src_imgs"]
    torch.cuda.synchronize()
    batch["src_imgs"] = torch.full(shape, 0.5, device=DEV)
    b = net(batch, False, False, 0.0, 0.0, out_depth=True, chunk=32)[1][0]
    assert enc.calls == 2 and max_abs(b, ref) > 1e-3             # really rendered from the new features
    # in-place edit of the same tensor
    batch["src_imgs"].fill_(0.0)
    c = net(batch, False, False, 0.0, 0.0, out_depth=True, chunk=32)[1][0]
    assert enc.calls == 3 and max_abs(c, ref) < 1e-6
    # unchanged inputs: cached
    net(batch, False, False, 0.0, 0.0, out_depth=True, chunk=32)
    assert enc.calls == 3
    # encoder weights change (fine-tuning / load_state_dict)
    with torch.no_grad():
        enc.gain.mul_(1.5)
    d = net(batch, False, False, 0.0, 0.0, out_depth=True, chunk=32)[1][0]
    assert enc.calls == 4 and max_abs(d, ref) > 1e-3


def test_training_step_between_two_eval_frames_does_not_leave_a_stale_scene():
    """An attached encoder, an eval frame of scene B, a DIFFERENTIABLE training call on scene A (which uploads A's encoder output
    through gather_features), then scene B again: the second eval frame must be rendered from B's features, not from what the
    training step left on the device (the eval-side cache key is dropped by every upload)."""
    enc = _StubEncoder().to(DEV)
    net = models.NeRF_TP(num_coarse_samples=16, num_fine_samples=24, num_src_views=cases.NV, encoder=enc).to(DEV)
    net.load_state_dict(synth.nerf_tp_state(0), strict=False)
    eval_b = _batch(48)                                                    # src_imgs = 0 -> k = 1
    train_a = dict(_batch(32))
    train_a["src_imgs"] = torch.full_like(eval_b["src_imgs"], 0.7)         # another scene
    first = net(eval_b, False, False, 0.0, 0.0, out_depth=True)[1][0].clone()
    calls = enc.calls
    with torch.enable_grad():
        for p in net.parameters():
            p.requires_grad_(True)
        out = net(train_a, True, False, 0.0, 0.0, out_depth=False, seed=5)
        assert out[1][0].requires_grad and enc.calls == calls + 1          # the encoder ran WITH autograd for the training batch
        out[1][0].sum().backward()
        assert enc.gain.grad is not None and float(enc.gain.grad.abs()) > 0.0      # gradients reach the encoder through the lookups
    for p in net.parameters():
        p.requires_grad_(False)
    enc.gain.grad = None
    again = net(eval_b, False, False, 0.0, 0.0, out_depth=True)[1][0]
    assert enc.calls == calls + 2                                          # re-encoded: the device scene was the training batch's
    assert max_abs(again, first) < 1e-6


def test_sphere_miss_raises():
    net = _net(32, 64)
    batch = _batch(8)
    batch["rays_o"] = batch["rays_o"].clone()
    batch["rays_o"][3] = torch.tensor([0.0, 0.0, 5.0], device=DEV)
    batch["rays_d"] = batch["rays_d"].clone()
    batch["rays_d"][3] = torch.tensor([1.0, 0.0, 0.0], device=DEV)
    with pytest.raises(AssertionError):              # default: deferred read, raised by check_flags()
        net(batch, False, False, 0.0, 0.0, out_depth=True)
        net.check_flags()
    net.poll_flags = "immediate"                     # the reference's behaviour: the call itself raises
    with pytest.raises(AssertionError):
        net(batch, False, False, 0.0, 0.0, out_depth=True)


def test_scene_required():
    net = models.NeRF_TP(num_coarse_samples=32, num_fine_samples=64).to(DEV)
    from neo360_amd import _lib
    with pytest.raises(_lib.NeoError):
        net(_batch(4), False, False, 0.0, 0.0, out_depth=True)
