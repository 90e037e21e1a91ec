"""GPU: round-6 host-side behaviour of the drop-in modules.

* consecutive evaluation calls of NeRF_TP overlap on the device (context.CallOverlap: two side streams x two scratch lanes of
  the context) - the reference's own chunk loop, 300 forward calls of 1024 rays (neo360/model.py:861-907) - while the
  caller's stream semantics stay those of a single-stream call;
* the advisor's round-5 findings (gather_map geometry check, forward-only callers outside no_grad).
"""
import pytest
import torch

import cases
from conftest import max_abs
from neo360_amd import _lib, models, ops, render, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
PER_RAY = ("rays_o", "rays_d", "viewdirs")


def _tp_net(scene, n_coarse=16, n_fine=32, **kw):
    net = models.NeRF_TP(num_coarse_samples=n_coarse, num_fine_samples=n_fine, num_src_views=cases.NV, **kw).to(DEV)
    net.load_state_dict(synth.nerf_tp_state(0))
    net.set_scene(scene["plane_xz"].to(DEV), scene["plane_xy"].to(DEV), scene["plane_yz"].to(DEV), scene["latent"].to(DEV),
                  scene["image_wh"])
    return net


def _batch(n):
    return {k: v.to(DEV) for k, v in cases.neo_batch(cases.strided_rays(n)).items()}


def _chunk_loop(net, batch, chunk):
    """The reference's render_rays_test loop, verbatim in shape: slices per-ray keys, passes src_* whole, cats at the end."""
    n = batch["rays_o"].shape[0]
    rgb, depth = [], []
    for i in range(0, n, chunk):
        part = {k: (v[i:i + chunk] if k in PER_RAY else v) for k, v in batch.items()}
        res = net(part, False, False, 0.0, 0.0, out_depth=True)
        rgb.append(res[1][0])
        depth.append(res[1][5])
    return torch.cat(rgb), torch.cat(depth)


def test_chunk_loop_frame_is_bitwise_the_whole_frame_call():
    """300-call-style loop (overlapped: lanes + side streams) == ONE library call with the chunk passed down == the same loop
    with overlap switched off.  Bitwise: the same kernels on the same operands, whatever runs concurrently."""
    sc = cases.small_scene()
    net = _tp_net(sc)
    batch = _batch(2348)                      # 1024 + 1024 + a short 300
    whole = render.render_rays_test(net, batch, chunk=1024)
    ctx = net._context(torch.device(DEV))
    ov = net._overlap(torch.device(DEV))
    calls0, fresh0 = ov.calls, ov.fresh_forks
    rgb, depth = _chunk_loop(net, batch, 1024)
    net.check_flags()
    assert ov.calls - calls0 == 3
    assert ov.fresh_forks - fresh0 <= 1, "calls 2 and 3 slice the same ray tensors: they reuse call 1's fork point"
    assert torch.equal(rgb, whole["rgb"]) and torch.equal(depth, whole["depth"])
    net.overlap_calls = False
    rgb2, depth2 = _chunk_loop(net, batch, 1024)
    net.check_flags()
    assert torch.equal(rgb2, rgb) and torch.equal(depth2, depth)
    assert ctx.sync_count() >= 0


def test_overlapped_calls_keep_the_callers_stream_semantics():
    """Outputs are consumed on the caller's stream straight after each call (no synchronisation), inputs are produced on it
    just before (a fresh tensor per call: every call forks from a fresh event), and freed output blocks are recycled while
    later calls are still in flight.  Many short calls; every result must equal the one-call-at-a-time result."""
    sc = cases.small_scene()
    net = _tp_net(sc)
    base = _batch(256 * 12)
    want = []
    net.overlap_calls = False
    for i in range(12):
        part = {k: (v[256 * i:256 * (i + 1)] if k in PER_RAY else v) for k, v in base.items()}
        want.append(net(part, False, False, 0.0, 0.0, out_depth=True)[1][0].clone())
    net.check_flags()
    net.overlap_calls = True
    ov = net._overlap(torch.device(DEV))
    fresh0 = ov.fresh_forks
    acc = []
    for rep in range(3):
        for i in range(12):
            # inputs made on the caller's stream right before the call (copies: new tensor objects every time)
            part = {k: ((v[256 * i:256 * (i + 1)] * 1.0) if k in PER_RAY else v) for k, v in base.items()}
            res = net(part, False, False, 0.0, 0.0, out_depth=True)
            acc.append(res[1][0] * 1.0)          # consumed on the caller's stream, no sync; `res` is freed right here
            del res, part
    net.check_flags()
    torch.cuda.synchronize()
    assert ov.fresh_forks - fresh0 == 36, "fresh input tensors: no fork point may be reused"
    for j, got in enumerate(acc):
        assert torch.equal(got, want[j % 12]), j


def test_fork_point_is_not_reused_across_rewritten_rays():
    """Same tensor objects, new contents: an in-place torch write bumps the version, an `out=` write of this library bumps
    the library's write epoch - either way the next call takes a fresh fork event (and sees the new rays)."""
    sc = cases.small_scene()
    net = _tp_net(sc)
    b = _batch(512)
    ov = net._overlap(torch.device(DEV))
    r1 = net(b, False, False, 0.0, 0.0, out_depth=True)[1][0].clone()
    f0 = ov.fresh_forks
    r1b = net(b, False, False, 0.0, 0.0, out_depth=True)[1][0].clone()
    assert ov.fresh_forks == f0 and torch.equal(r1, r1b)
    b["viewdirs"].copy_(torch.nn.functional.normalize(b["viewdirs"] + 0.25, dim=-1))     # in place: version + 1
    r2 = net(b, False, False, 0.0, 0.0, out_depth=True)[1][0].clone()
    assert ov.fresh_forks == f0 + 1 and not torch.equal(r1, r2)
    e0 = _lib.write_epoch
    H = W = 16
    rays = ops.get_ray_directions_and_rays(H, W, 12.8, synth.look_at_origin(40.0))
    ops.get_ray_directions_and_rays(H, W, 12.8, synth.look_at_origin(75.0), out=rays)
    assert _lib.write_epoch == e0 + 1
    net.check_flags()


def test_uploads_stay_ordered_against_both_lanes():
    """A weight change between two overlapped calls: the upload is an exclusive call of the context (waits for both lanes, and
    both lanes wait for it) - the call before it sees the old weights, the call after it the new ones."""
    sc = cases.small_scene()
    net = _tp_net(sc)
    b = _batch(1024)
    net.overlap_calls = False
    old = net(b, False, False, 0.0, 0.0, out_depth=True)[1][0].clone()
    st = synth.nerf_tp_state(0)
    st2 = {k: (v * 1.25 if "rgb_layer.weight" in k else v) for k, v in st.items()}
    net.load_state_dict(st2)
    new = net(b, False, False, 0.0, 0.0, out_depth=True)[1][0].clone()
    net.check_flags()
    assert not torch.equal(old, new)
    net.overlap_calls = True
    for _ in range(4):
        net.load_state_dict(st)
        a1 = net(b, False, False, 0.0, 0.0, out_depth=True)[1][0]
        a2 = net(b, False, False, 0.0, 0.0, out_depth=True)[1][0]
        net.load_state_dict(st2)
        c1 = net(b, False, False, 0.0, 0.0, out_depth=True)[1][0]
        c2 = net(b, False, False, 0.0, 0.0, out_depth=True)[1][0]
        net.check_flags()
        assert torch.equal(a1, old) and torch.equal(a2, old) and torch.equal(c1, new) and torch.equal(c2, new)


# ---- advisor, round 5 ---------------------------------------------------------------------------------------------------------
def test_gather_map_rejects_a_map_of_another_geometry():
    """training.gather_map indexes the caller's map from the geometry uploaded in the module's context: a map with another row
    count (stale scene, another resolution) used to be read - and scattered into - out of bounds; now the library refuses it."""
    from neo360_amd import training
    sc = cases.small_scene()
    net = _tp_net(sc)
    b = _batch(64)
    nv, _, hf, wf = sc["latent"].shape
    pts = (torch.rand(200, 3, device=DEV) - 0.5)
    with torch.enable_grad():
        good = torch.randn(nv * hf * wf, 64, device=DEV, requires_grad=True)
        out = training.gather_map(net, good, pts, b)
        out.sum().backward()
    assert good.grad is not None and good.grad.shape == good.shape
    for rows in (nv * hf * wf - wf, nv * hf * wf + 1, nv * (hf // 2) * (wf // 2)):
        bad = torch.randn(rows, 64, device=DEV)
        with pytest.raises(_lib.NeoError, match="map rows differ"):
            training.gather_map(net, bad, pts, b)


def test_forward_only_caller_outside_no_grad_is_told_once_and_chunked_pixelnerf_still_works():
    """parameters default to requires_grad=True, so a deterministic forward outside no_grad lands on the operator
    chain by the automatic rule.  It says so once per module; and where only the fused path can serve the call (PixelNeRF with
    `chunk`, a set_scene latent the caller dropped) the automatic rule falls back to it instead of raising."""
    import warnings
    sc = cases.small_scene()
    pix = models.PixelNeRF(num_src_views=cases.NV).to(DEV)
    pix.load_state_dict(synth.pixelnerf_state(0))
    pix.set_scene(sc["latent"].to(DEV), sc["image_wh"])          # the caller keeps no reference: the module's is weak
    b = _batch(96)
    with torch.no_grad():
        want = pix(b, False, False, 0.2, 3.0, chunk=32)
    with warnings.catch_warnings(record=True) as seen, torch.enable_grad():
        warnings.simplefilter("always")
        got = pix(b, False, False, 0.2, 3.0, chunk=32)            # grad mode on, deterministic, chunked: fused fallback
        got2 = pix(b, False, False, 0.2, 3.0)                     # latent tensor is gone: fused fallback as well
    assert torch.equal(got[1][0], want[1][0])
    assert got2[1][0].shape == want[1][0].shape and not got2[1][0].requires_grad
    assert sum("differentiable operator chain" in str(w.message) for w in seen) == 1
    pix.differentiable = True                                      # an explicit request keeps the strict behaviour
    with pytest.raises(NotImplementedError), torch.enable_grad():
        pix(b, False, False, 0.2, 3.0, chunk=32)


def test_ray_patch_order_is_bitwise_neutral():
    """The pixel-grid hint of the frame API (neo_ctx_set_ray_grid: rays walked in 8 x 8 pixel patches inside whole bands of 8
    image rows) is pure scheduling: the frame, and any shard of it - band-aligned or with ragged ends - is bitwise the frame
    rendered in the caller's ray order."""
    sc = cases.small_scene()
    net = _tp_net(sc)
    Hs, Ws = 48, 64
    ro, vd, rd, _ = ops.get_ray_directions_and_rays(Hs, Ws, 0.8 * Ws, synth.look_at_origin(40.0))
    extra = {k: v for k, v in _batch(8).items() if k.startswith("src_")}
    frame = dict(rays_o=ro, rays_d=rd, viewdirs=vd, **extra)
    plain = render.render_rays_test(net, frame, chunk=1024)
    hinted = render.render_rays_test(net, frame, chunk=1024, image_width=Ws)
    assert torch.equal(plain["rgb"], hinted["rgb"]) and torch.equal(plain["depth"], hinted["depth"])
    ctx = net._context(torch.device(DEV))
    assert getattr(ctx, "_ray_grid", (0, 0)) == (0, 0), "the hint must not outlive the frame call"
    for lo, hi in ((1024, 3072), (300, 2348), (0, 1000)):
        # Q1 (view-direction tiling) depends on chunk membership: shards start on chunk boundaries in the real sharded render;
        # here the whole shard is ONE chunk on both sides, so any range is comparable
        part = {k: (v[lo:hi] if k in PER_RAY else v) for k, v in frame.items()}
        a = render.render_rays_test(net, part, chunk=hi - lo)
        b = render.render_rays_test(net, part, chunk=hi - lo, image_width=Ws, first_ray=lo)
        assert torch.equal(a["rgb"], b["rgb"]) and torch.equal(a["depth"], b["depth"]), (lo, hi)
    with pytest.raises(_lib.NeoError):
        ctx.set_ray_grid(60)                 # not a multiple of 8


@pytest.mark.parametrize("which", ["vanilla", "pixelnerf", "mip360"])
def test_other_renderers_chunk_loops_overlap_and_stay_bitwise(which):
    """The reference drives every renderer chunk by chunk (vanilla_nerf/model.py:336-363, model_pixel.py:356-383,
    mipnerf360/model.py:471-505).  Their fused evaluation calls take the same side-stream / scratch-lane overlap as NeRF_TP's:
    the overlapped loop's frame is bitwise the serial loop's."""
    n, chunk = 1536, 256
    if which == "vanilla":
        net = models.NeRF().to(DEV)
        net.load_state_dict(synth.vanilla_state(0))
        batch = {k: v.to(DEV) for k, v in cases.strided_rays(n).items()}
        call = lambda part: net(part, False, False, 0.2, 3.0)[1][0]
    elif which == "pixelnerf":
        sc = cases.small_scene()
        net = models.PixelNeRF(num_src_views=cases.NV).to(DEV)
        net.load_state_dict(synth.pixelnerf_state(0))
        latent = sc["latent"].to(DEV)
        net.set_scene(latent, sc["image_wh"])
        batch = _batch(n)
        call = lambda part: net(part, False, False, 0.2, 3.0)[1][0]
    else:
        net = models.MipNeRF360(num_prop_samples=16, num_nerf_samples=8).to(DEV)
        net.load_state_dict(synth.mip360_state(0, weight_gain=0.5))
        batch = {k: v.to(DEV) for k, v in cases.mip_rays(n).items()}
        call = lambda part: net(part, 1.0, False, False, 0.2, 3.0)[0][-1]["rgb"]
    per_ray = PER_RAY + ("radii",)

    def loop():
        out = []
        for i in range(0, n, chunk):
            part = {k: (v[i:i + chunk] if k in per_ray else v) for k, v in batch.items()}
            out.append(call(part))
        res = torch.cat(out)
        net.check_flags()
        return res
    net.overlap_calls = False
    want = loop()
    net.overlap_calls = True
    ov = net._overlap(torch.device(DEV))
    c0, f0 = ov.calls, ov.fresh_forks
    got = loop()
    assert ov.calls - c0 == n // chunk and ov.fresh_forks - f0 <= 1
    assert torch.equal(got, want)
    got2 = loop()
    assert torch.equal(got2, want)


@pytest.mark.parametrize("rows_p,nv,input_ch", [(1000, 3, 3), (37, 2, 3), (4133, 3, 3), (777, 3, 4)])
def test_fused_training_chain_equals_the_layer_by_layer_chain(rows_p, nv, input_ch):
    """neo_train_chain_mode: the per-row part of the projected-space NeRFPPMLP chain as ONE kernel each way (csrc/train_chain.h: the
    activation tile stays in LDS, every layer goes to HBM once; bottleneck and view layer 0 on the view MEANS - P rows instead of NV P,
    by linearity) against one GEMM launch per layer in the reference's order: outputs, all 18 parameter gradients, g_world and g_pre
    agree to rounding of the summation order; ragged last tiles (rows not a multiple of 64) and tiles that straddle two views
    included."""
    from neo360_amd import training
    lib = _lib.load()
    mlp = models.NeRFPPMLP(0, 10, 4, input_ch=input_ch, num_src_views=nv).to(DEV)              # input_ch 4: the outside-sphere MLPs (84-d encoding)
    g = torch.Generator(device=DEV).manual_seed(rows_p)
    with torch.no_grad():
        for p in mlp.parameters():
            p.copy_(torch.randn(p.shape, device=DEV, generator=g) * (0.15 if p.dim() == 2 else 0.05))
    P = rows_p
    x_enc = torch.randn(nv, P, 21 * input_ch, device=DEV, generator=g)
    cond = torch.randn(nv * P, 27, device=DEV, generator=g)
    world0 = torch.randn(nv * P, 128, device=DEV, generator=g) * 0.5
    pre0 = torch.randn(nv * P, 256, device=DEV, generator=g) * 0.5
    up_rgb, up_sigma = torch.randn(P, 3, device=DEV, generator=g), torch.randn(P, 1, device=DEV, generator=g)
    layers = mlp.ordered_layers()
    params = [l.weight for l in layers] + [l.bias for l in layers]
    old = lib.neo_train_chain_mode(-1)
    res = {}
    try:
        with torch.enable_grad():
            for p in params:
                p.requires_grad_(True)
            for mode in (1, 0):
                lib.neo_train_chain_mode(mode)
                world, pre = world0.clone().requires_grad_(True), pre0.clone().requires_grad_(True)
                rgb, sigma = training.nerfpp_mlp_projected(mlp, x_enc, cond, world, pre, nv)
                grads = torch.autograd.grad((rgb * up_rgb).sum() + (sigma * up_sigma).sum(), [world, pre] + params)
                res[mode] = (rgb.detach(), sigma.detach(), grads)
    finally:
        lib.neo_train_chain_mode(old)
    (rgb_a, sig_a, g_a), (rgb_b, sig_b, g_b) = res[1], res[0]
    assert max_abs(rgb_a, rgb_b) <= 5e-6 * max(1.0, float(rgb_b.abs().max())) and max_abs(sig_a, sig_b) <= 5e-6 * max(1.0, float(sig_b.abs().max()))
    # A unit whose pre-activation rounds to +0 in one schedule and to a tiny positive number in the other takes the other ReLU derivative
    # (about one unit in 6 M at the largest size): its ROW's input gradients then differ by a finite amount - a property of ReLU under any
    # change of summation order, not of a schedule.  Rows are therefore compared one by one, a handful may differ, and the parameter
    # gradients (sums over all rows) are held to the tight bound only when no row flipped.
    flipped = 0
    for a, b in zip(g_a[:2], g_b[:2]):
        bad = ((a - b).abs().amax(dim=1) > 1e-5 * max(float(b.abs().max()), 1e-6))
        flipped = max(flipped, int(bad.sum()))
    assert flipped <= 3, flipped
    for i, (a, b) in enumerate(zip(g_a[2:], g_b[2:])):
        err, ref = float((a - b).abs().max()), max(float(b.abs().max()), 1e-6)
        if flipped == 0:
            assert err <= 2e-5 * ref, (i, err, ref)
        else:
            assert float((a - b).norm()) <= 5e-3 * float(b.norm()) + 1e-12, (i, err, ref, flipped)


def test_merged_projection_shares_one_gradient_buffer_and_survives_a_second_backward():
    """project_latent_all: one texel-space GEMM for the four MLPs, each lookup reads / scatters into its 256-column slice
    (neo_tp_gather_map_slice) of ONE map / ONE gradient buffer.  Against one projection + one full-size gradient per MLP: same values,
    same gradients; and the shared buffer is released after a backward pass, so a second pass over a retained graph gives the same
    gradients again (not twice the first)."""
    from neo360_amd import training
    sc = cases.small_scene()
    net = _tp_net(sc)
    batch = _batch(64)
    mlps = net._mlps()
    g = torch.Generator(device=DEV).manual_seed(21)
    latent = sc["latent"].to(DEV)
    lat_cl = latent.permute(0, 2, 3, 1).reshape(-1, latent.shape[1]).contiguous().requires_grad_(True)
    pts = (torch.rand(500, 3, device=DEV, generator=g) - 0.5) * 1.2
    ups = [torch.randn(cases.NV * 500, 256, device=DEV, generator=g) for _ in mlps]
    with torch.enable_grad():
        for p in net.parameters():
            p.requires_grad_(True)
        gmap, shared = training.project_latent_all(mlps, lat_cl)
        outs = [training.gather_map(net, gmap, pts + 0.01 * i, batch, col=256 * i, width=256, shared=shared) for i in range(4)]
        loss = sum((o * u).sum() for o, u in zip(outs, ups))
        wanted = [lat_cl] + [m.pts_linears[j].weight for m in mlps for j in (0, 3)]
        g1 = torch.autograd.grad(loss, wanted, retain_graph=True)
        g2 = torch.autograd.grad(loss, wanted)
        singles = [training.gather_map(net, training.project_latent(m, lat_cl), pts + 0.01 * i, batch) for i, m in enumerate(mlps)]
        g3 = torch.autograd.grad(sum((o * u).sum() for o, u in zip(singles, ups)), wanted)
    assert shared["buf"] is None
    for o, s in zip(outs, singles):
        assert max_abs(o.detach(), s.detach()) <= 1e-6 * max(1.0, float(s.abs().max()))
    for a, b, c in zip(g1, g2, g3):
        assert float((a - b).abs().max()) <= 2e-6 * max(float(b.abs().max()), 1e-6)          # atomics: the order of the adds differs
        assert float((a - c).abs().max()) <= 2e-5 * max(float(c.abs().max()), 1e-6)


@pytest.mark.parametrize("shape", [(3, 512, 24, 40), (2, 130, 7, 9), (1, 64, 1, 1)])
def test_channels_last_rows_is_permute_reshape_both_ways(shape):
    """training.channels_last_rows (neo_transpose, a tiled transpose each way) = permute(0, 2, 3, 1).reshape(-1, C) under autograd,
    bitwise (a copy), ragged tiles included."""
    from neo360_amd import training
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = torch.randn(*shape, device=DEV, generator=g).requires_grad_(True)
    up = torch.randn(shape[0] * shape[2] * shape[3], shape[1], device=DEV, generator=g)
    with torch.enable_grad():
        y = training.channels_last_rows(x)
        want = x.permute(0, 2, 3, 1).reshape(-1, shape[1])
        assert torch.equal(y.detach(), want.detach())
        (ga,) = torch.autograd.grad((y * up).sum(), [x])
        (gb,) = torch.autograd.grad((want * up).sum(), [x])
    assert torch.equal(ga, gb)


def test_shared_gradient_buffers_across_chunks_and_partial_losses():
    """The lookups of one training call scatter into shared gradient buffers (one per tri-plane, one for the merged projected map) whatever
    the chunking; a loss that reads only the fine level (the coarse lookups never run their backward) still gets the complete
    gradient.  Against the per-row path (no projection, no sharing), which the fp64-autograd tests pin to the oracle."""
    sc = cases.small_scene()
    R = 96
    gb = _batch(R)
    target = synth.uniform(23, "chunk_target", (R, 3), 0.0, 1.0).to(DEV)
    results = {}
    # (the reference takes the same chunk: quirk Q1 makes the direction encodings of a call depend on its chunk size)
    for tag, projected, chunk, fine_only in (("ref", False, 32, False), ("chunked", True, 32, False), ("ref_fine", False, 32, True),
                                             ("chunked_fine", True, 32, True), ("private", True, 32, False)):
        net = models.NeRF_TP(num_coarse_samples=16, num_fine_samples=24, num_src_views=cases.NV).to(DEV)
        net.load_state_dict(synth.nerf_tp_state(0))
        net.train_projected = projected
        net.train_shared_grads = tag != "private"          # "private": one gradient buffer per lookup, summed by autograd
        maps = [sc[k].to(DEV).clone().requires_grad_(True) for k in ("plane_xz", "plane_xy", "plane_yz", "latent")]
        with torch.enable_grad():
            net.set_scene(*maps, sc["image_wh"])
            for p in net.parameters():
                p.requires_grad_(True)
            kw = {} if chunk is None else {"chunk": chunk}
            out = net(gb, True, False, 0.0, 0.0, out_depth=False, seed=5, **kw)
            levels = out[1:] if fine_only else out
            loss = sum(((lv[0] - target) ** 2).mean() for lv in levels)
            grads = torch.autograd.grad(loss, maps, allow_unused=True)
        results[tag] = [g.detach() for g in grads]
    for a_tag, b_tag in (("ref", "chunked"), ("ref_fine", "chunked_fine"), ("ref", "private")):
        for nm, a, b in zip(("plane_xz", "plane_xy", "plane_yz", "latent"), results[a_tag], results[b_tag]):
            rel = float((a - b).norm()) / (float(a.norm()) + 1e-20)
            assert rel < 2e-3 and float(b.abs().max()) > 0.0, (a_tag, nm, rel)
