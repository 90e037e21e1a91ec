"""GPU: the BASELINE.json configurations at their FULL sizes inside the -m gpu suite: NeO-360 640x480 with
3 x 128 x 120 x 160 tri-planes and 3 x 512 x 240 x 320 latents (C3), Mip-NeRF 360 640x480 (C5).  Whole frames are
checked through size-independent properties (finite, ranges, partition of opacity, row-order independence, chunk
structure).  ONE REFERENCE CHUNK of the C3 configuration (1024 rays of the bench frame, full-size features from the
hash generator) is compared with the REFERENCE ITSELF: fixture g4_neo_full = models/neo360/model.py:266-581 run in the
build container at num_coarse_samples=128, num_fine_samples=256 (:169-171), chunk 1024 (opt.py:195-200);
g4_neo_full_noise = its fp64 twin (tests/golden/make_golden.py)."""
import pytest
import torch

import cases
from conftest import check_vs_reference_noise, max_abs, record_parity
from neo360_amd import models, ops, render, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
H, W = 480, 640
NV = 3


@pytest.fixture(scope="module")
def neo_full():
    state = synth.nerf_tp_state(0)
    net = models.NeRF_TP(num_coarse_samples=128, num_fine_samples=256, num_src_views=NV).to(DEV)
    net.load_state_dict(state)
    # the scene the build container ran the reference on (cases.full_scene: hash generator, bit-identical anywhere)
    scene = {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in cases.full_scene().items()}
    net.set_scene(scene["plane_xz"], scene["plane_xy"], scene["plane_yz"], scene["latent"], scene["image_wh"])
    poses, focal, centre = synth.source_views(NV, W, H)
    ro, vd, rd, radii = ops.get_ray_directions_and_rays(H, W, 0.8 * W, synth.look_at_origin(40.0))
    batch = dict(rays_o=ro, rays_d=rd, viewdirs=vd, src_poses=poses.to(DEV), src_focal=focal.to(DEV),
                 src_c=centre.to(DEV), src_imgs=torch.zeros(NV, 3, H, W, device=DEV))
    return state, net, scene, batch


def test_neo360_full_frame_properties(neo_full):
    state, net, scene, batch = neo_full
    res = net(batch, False, False, 0.0, 0.0, out_depth=True, chunk=1024)
    R = H * W
    for lv in (0, 1):
        rgb, fg, bg, acc, lam, depth = res[lv]
        assert rgb.shape == (R, 3) and acc.shape == (R,) and lam.shape == (R, 1) and depth.shape == (R,)
        for x in (rgb, fg, bg, acc, lam, depth):
            assert bool(torch.isfinite(x).all())
        # colours are convex combinations of sigmoid outputs in [-0.001, 1.001]
        assert float(rgb.min()) >= -2e-3 and float(rgb.max()) <= 1.002 + 1e-3
        assert float(acc.min()) >= 0.0 and float(acc.max()) <= 1.0 + 1e-5
        assert float(lam.min()) >= 0.0 and float(lam.max()) <= 1.0 + 1e-5
        # opacity inside the sphere + transmittance behind it partition 1 (helper.py:150-160: acc = 1 - T_last up to 1e-10 terms)
        assert max_abs(acc + lam.squeeze(-1), torch.ones(R)) < 1e-4
        assert max_abs(rgb, fg + lam * bg) < 1e-6
        assert float(depth.min()) >= 0.0
    # chunk structure: reference chunks 64..68 of the frame (rays 65536..70655) rendered alone give the same pixels
    lo, hi = 64 * 1024, 69 * 1024
    sub = {k: (v if k.startswith("src_") else v[lo:hi]) for k, v in batch.items()}
    part = net(sub, False, False, 0.0, 0.0, out_depth=True, chunk=1024)
    assert torch.equal(part[1][0], res[1][0][lo:hi]) and torch.equal(part[1][5], res[1][5][lo:hi])
    # whole-frame API == the module call
    out = render.render_rays_test(net, batch, chunk=1024)
    assert torch.equal(out["rgb"], res[1][0]) and torch.equal(out["depth"], res[1][5])


def _outputs(res):
    cat = lambda lv, j: res[lv][j].cpu()
    return dict(rgb0=cat(0, 0), depth0=cat(0, 5), rgb1=cat(1, 0), fg1=cat(1, 1), bg1=cat(1, 2), fgacc1=cat(1, 3),
                lam1=cat(1, 4), depth1=cat(1, 5))


_NV5 = {}


def _net_nv5():
    """Five source views at full map size (fixture g4_neo_full_b4; the reference builds NeRF_TP(num_src_views=5) for the
    5-view render names, neo360/model.py:606-616)."""
    if "net" not in _NV5:
        net = models.NeRF_TP(num_coarse_samples=128, num_fine_samples=256, num_src_views=5).to(DEV)
        net.load_state_dict(synth.nerf_tp_state(0))
        sc = cases.full_scene(nv=5)
        net.set_scene(sc["plane_xz"].to(DEV), sc["plane_xy"].to(DEV), sc["plane_yz"].to(DEV), sc["latent"].to(DEV), sc["image_wh"])
        _NV5["net"] = net
    return _NV5["net"]


_VARIANT = {}


def _net_variant(tag):
    """Fixtures b5 / b6 (round 5): their own weights (density gain 8: trained-like sharp densities) / their own scene (feature
    seed 1 at std 0.5), cases.FULL_B."""
    if tag not in _VARIANT:
        net = models.NeRF_TP(num_coarse_samples=128, num_fine_samples=256, num_src_views=NV).to(DEV)
        net.load_state_dict(synth.nerf_tp_state(0, density_gain=cases.full_gain(tag)))
        sc, _ = cases.full_case(tag, 8)
        net.set_scene(sc["plane_xz"].to(DEV), sc["plane_xy"].to(DEV), sc["plane_yz"].to(DEV), sc["latent"].to(DEV), sc["image_wh"])
        _VARIANT[tag] = net
    return _VARIANT[tag]


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
@pytest.mark.parametrize("tag", ["", "b1", "b2", "b3", "b4", "b5", "b6"])
def test_neo360_full_size_chunk_vs_reference(neo_full, golden, golden_optional, precision, tag):
    """Reference chunks (1024 rays spread over the 640x480 frame) at the full C3 configuration against the reference's own
    outputs: the round-3 chunk (""), four more (cases.FULL_B: another strip, two other target poses, five source
    views) and - round 5 - b5 (density gain 8: trained-like) and b6 (a second feature seed at std 0.5).  Same rule as the small-scene 1024-ray test: 1e-4 on EVERY output (depth included) of every ray the reference
    determines to better than 1e-5; a ray the reference disagrees with itself on within 1e-4 + 3 x that disagreement; a
    flip-prone ray within 1e-4 + 3 x ITS OWN flip size (conftest.check_vs_reference_noise)."""
    state, net, scene, batch = neo_full
    name = "g4_neo_full" + ("_" + tag if tag else "")
    g, noise = golden(name), golden(name + "_noise")
    if tag == "b4":
        net = _net_nv5()
    elif tag in ("b5", "b6"):
        net = _net_variant(tag)
    _, cb = cases.full_case(tag) if tag else (None, cases.full_batch(1024))
    gb = {k: v.to(DEV) for k, v in cb.items()}
    old = net.precision
    net.precision = precision
    try:
        got = _outputs(net(gb, False, False, 0.0, 0.0, out_depth=True))
        net.check_flags()
    finally:
        net.precision = old
    check_vs_reference_noise(got, g, noise, "neo360_full_size_C3%s/%s" % ("_" + tag if tag else "", precision),
                             flip=golden_optional(name + "_flip"))
    if not tag:
        # the library's ray generator gives the same rays as the fixture's fp64 NumPy ones (o exact, d to fp32 rounding):
        # the chunk is a subset of the frame the benchmark renders
        idx = cases.full_strip_index(1024).to(DEV)
        assert torch.equal(batch["rays_o"][idx], gb["rays_o"]) and max_abs(batch["rays_d"][idx], gb["rays_d"]) < 3e-7


@pytest.mark.parametrize("tag", ["", "b5"])
def test_neo360_full_size_every_ray_at_the_gpus_own_positions(neo_full, tag):
    """The 1e-4 contract on EVERY ray of a full-size C3 chunk, no margin / flip / distribution rule (VERDICT r5 task 5).
    The rules of test_neo360_full_size_chunk_vs_reference exist because the reference's level-1 POSITIONS are ill-conditioned
    (an ulp of the coarse cdf moves a sample across a bin; its own fp32 and fp64 runs disagree on b5 by more than 1e-4).  Here
    the positions are taken out of the comparison: the GPU renders, exposes the level-1 positions it used
    (`sample_positions`), and the pinned oracle (tests/test_oracle_fullsize.py: == the reference <= 5e-6 at this size)
    evaluates the reference's fp32 arithmetic AT those positions on the host.  Everything else - lookups, encodings, the four
    MLPs, the view-direction tiling, compositing, merge - is compared at 1e-4 on all eight outputs of all 1024 rays, for the
    bench chunk ("") and for the trained-like one (b5, density gain 8)."""
    import oracle
    state, net, scene, batch = neo_full
    if tag:
        net = _net_variant(tag)
    sc_cpu, cb = cases.full_case(tag, 1024)
    gb = {k: v.to(DEV) for k, v in cb.items()}
    res = net(gb, False, False, 0.0, 0.0, out_depth=True)
    pos = net.sample_positions(gb)
    net.check_flags()
    got = _outputs(res)
    # the positions really are those of the evaluation call: its level-1 colour is the training-shaped call's, bitwise
    again = net(gb, False, False, 0.0, 0.0, out_depth=False)
    assert torch.equal(again[1][0], res[1][0]) and torch.equal(again[0][0], res[0][0])
    fg_t1, bg_s1 = pos[1][0].cpu(), pos[1][1].cpu()
    assert fg_t1.shape == (1024, 385) and bool((fg_t1[:, 1:] >= fg_t1[:, :-1]).all()) and bool((bg_s1[:, 1:] <= bg_s1[:, :-1]).all())
    params = synth.nerf_tp_state(0, density_gain=cases.full_gain(tag))
    torch.set_num_threads(max(1, min(64, (torch.get_num_threads() or 1))))
    want = oracle.neo360.render(params, cb, sc_cpu, 128, 256, fine_samples=(fg_t1, bg_s1))
    want = dict(rgb0=want[0][0], depth0=want[0][5], rgb1=want[1][0], fg1=want[1][1], bg1=want[1][2], fgacc1=want[1][3],
                lam1=want[1][4], depth1=want[1][5])
    worst = {}
    for k in want:
        e = (got[k].double() - want[k].double()).abs()
        e = e.reshape(e.shape[0], -1).amax(dim=1)
        worst[k] = (float(e.max()), int((e >= 1e-4).sum()))
    record_parity("neo360_full_size_C3%s/every_ray_at_gpu_positions" % ("_" + tag if tag else ""),
                  rays=1024, **{"max_" + k: v[0] for k, v in worst.items()}, rays_above_1e4=sum(v[1] for v in worst.values()))
    for k, (mx, n) in worst.items():
        assert n == 0 and mx < 1e-4, (tag, k, mx, n)


def test_mip360_full_frame_and_strip():
    state = synth.mip360_state(0, weight_gain=0.5)
    net = models.MipNeRF360(num_prop_samples=64, num_nerf_samples=32).to(DEV)
    net.load_state_dict(state)
    ro, vd, rd, radii = ops.get_ray_directions_and_rays(H, W, 0.8 * W, synth.look_at_origin(40.0))
    batch = dict(rays_o=ro, rays_d=rd, viewdirs=vd, radii=radii[:, None])
    rend, hist = net(batch, 1.0, False, False, 0.2, 3.0)
    R = H * W
    for lv, n in enumerate((64, 64, 32)):
        rgb, w, sd = rend[lv]["rgb"], hist[lv]["weights"], hist[lv]["sdist"]
        assert rgb.shape == (R, 3) and w.shape == (R, n) and sd.shape == (R, n + 1)
        assert bool(torch.isfinite(rgb).all()) and bool(torch.isfinite(w).all())
        assert float(rgb.min()) >= -2e-3 and float(rgb.max()) <= 1.0 + 2e-3
        assert float(w.min()) >= 0.0 and max_abs(w.sum(-1), torch.ones(R)) < 1e-4      # opaque background: weights sum to 1
        assert bool((sd[:, 1:] >= sd[:, :-1]).all()) and float(sd.min()) >= 0.0 and float(sd.max()) <= 1.0
    idx = (torch.arange(128, device=DEV) * 1201 + 200 * W) % R
    sub = {k: v[idx].contiguous() for k, v in batch.items()}
    got, _ = net(sub, 1.0, False, False, 0.2, 3.0)
    torch.set_num_threads(32)
    from oracle import mip360 as oracle_mip360
    want, _ = oracle_mip360.render(state, {k: v.cpu() for k, v in sub.items()}, 1.0, 0.2, 3.0, num_prop_samples=64, num_nerf_samples=32)
    assert torch.equal(got[-1]["rgb"], rend[-1]["rgb"][idx])                            # rays are independent
    err = max_abs(got[-1]["rgb"], want[-1]["rgb"])
    record_parity("mip360_full_size_strip_vs_oracle", max_rgb=err, rays=128)
    assert err < 1e-4
