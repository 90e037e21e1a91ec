"""GPU: host-side behaviours added in round 5 (VERDICT r4 tasks 2, 5, 7).

* uploads / set_scene are ORDERED across streams like the renders: weights or the scene change on
  stream B while a frame is in flight on stream A.
* `NeRF.forward(randomized=True)` and the differentiable vanilla call (vanilla_nerf/model.py:154-216, :281-300): forward
  values and the gradients of the reference's loss against fp64 autograd of the oracle; `noise_std` / `density_noise`.
* the exact-fp32 PixelNeRF evaluator (`precision = "f32"`, vanilla_nerf/model_pixel.py:174-258) and the range-guard
  retry that lands on it.
* a static-operand range trip latches the module to the exact kernels until weights or scene change.
* `tp_render_train(chunk=...)` draws the same samples as the un-chunked call for one seed.
* `module.differentiable` makes the path choice explicit.
"""
import json
import os
import subprocess
import sys
import warnings

import pytest
import torch

import cases
import oracle
from conftest import max_abs, record_parity
from neo360_amd import _lib, models, ops, render, synth, training

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PER_RAY = ("rays_o", "rays_d", "viewdirs")


def _tp_net(scene, **kw):
    net = models.NeRF_TP(num_coarse_samples=16, num_fine_samples=32, num_src_views=cases.NV, **kw).to(DEV)
    net.load_state_dict(synth.nerf_tp_state(0))
    net.set_scene(scene["plane_xz"].to(DEV), scene["plane_xy"].to(DEV), scene["plane_yz"].to(DEV), scene["latent"].to(DEV),
                  scene["image_wh"])
    return net


def _batch(n):
    return {k: v.to(DEV) for k, v in cases.neo_batch(cases.strided_rays(n)).items()}


# ---- cross-stream ordering of uploads and set_scene -----------------------------------------------------------------

def test_weights_and_scene_change_between_streams_are_ordered():
    """Frame on stream A, then NEW WEIGHTS and a frame on stream B, back to A ...: the repack of stream B's upload must wait -
    on the device - for the frame of stream A that still reads the old fragments (the ORDERED scope used to cover the
    renders only).  Then the same with the scene.  Each frame equals the one-stream result of ITS (weights, scene), bitwise."""
    sc = cases.small_scene()
    batch = _batch(2048)
    states = [synth.nerf_tp_state(0), {k: v * 1.07 for k, v in synth.nerf_tp_state(0).items()}]
    scenes = [sc, {k: (v * 0.8 if isinstance(v, torch.Tensor) else v) for k, v in sc.items()}]

    def set_scene(net, i):
        s = scenes[i]
        net.set_scene(s["plane_xz"].to(DEV), s["plane_xy"].to(DEV), s["plane_yz"].to(DEV), s["latent"].to(DEV), s["image_wh"])

    ref = _tp_net(sc)
    want = {}
    for w in (0, 1):
        for s in (0, 1):
            ref.load_state_dict(states[w])
            set_scene(ref, s)
            want[(w, s)] = ref(batch, False, False, 0.0, 0.0, out_depth=True)[1][0].clone()
    ref.check_flags()
    assert max_abs(want[(0, 0)], want[(1, 0)]) > 1e-4 and max_abs(want[(0, 0)], want[(0, 1)]) > 1e-4

    # two modules with their OWN (never modified) parameter tensors drive ONE library context: every switch between them is an
    # upload of the other's weights into the context's fragment buffers - on the other stream, with no host synchronisation
    nets = [_tp_net(sc), models.NeRF_TP(num_coarse_samples=16, num_fine_samples=32, num_src_views=cases.NV).to(DEV)]
    nets[1].load_state_dict(states[1])
    nets[1]._ctx_cache = nets[0]._ctx_cache
    nets[1]._scene_ctx = nets[0]._scene_ctx
    ctx = nets[0]._context(torch.device(DEV))
    assert nets[1]._context(torch.device(DEV)) is ctx
    before = ctx.stream_waits()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    outs = []
    for it in range(10):
        i = it & 1
        with torch.cuda.stream(streams[i]):
            outs.append(((i, 0), nets[i](batch, False, False, 0.0, 0.0, out_depth=True)[1][0]))
    torch.cuda.synchronize()
    for it in range(4):                                # the scene as well (set_scene waits for the stream it is called on)
        i = it & 1
        with torch.cuda.stream(streams[i]):
            set_scene(nets[0], i)
            nets[1]._scene_ctx = nets[0]._scene_ctx
            outs.append(((1 - i, i), nets[1 - i](batch, False, False, 0.0, 0.0, out_depth=True)[1][0]))
    torch.cuda.synchronize()
    for n in nets:
        n.check_flags()
    assert ctx.stream_waits() >= before + 10
    for key, got in outs:
        assert torch.equal(got, want[key]), key
    nets[1]._ctx_cache = {}                            # one owner closes the shared context


# ---- vanilla NeRF: randomized / differentiable call ----------------------------------------------------------------------

def _vanilla(noise_std=0.0, n0=16, n1=24):
    net = models.NeRF(num_coarse_samples=n0, num_fine_samples=n1, noise_std=noise_std).to(DEV)
    net.load_state_dict(synth.vanilla_state(0))
    return net


def test_vanilla_randomized_forward_matches_oracle_at_its_samples():
    """randomized=True draws stratified level-0 samples and random quantiles; the oracle evaluated AT those positions (they
    carry no gradient) reproduces every output; the same seed repeats bitwise, another seed does not."""
    R, n0, n1 = 96, 16, 24
    net = _vanilla(n0=n0, n1=n1)
    rays_c = cases.strided_rays(R)
    rays = {k: v.to(DEV) for k, v in rays_c.items()}
    out, ts = training.nerf_render_train(net, rays, True, False, 0.2, 3.0, seed=77, return_samples=True)
    again = net(rays, True, False, 0.2, 3.0, seed=77)
    other = net(rays, True, False, 0.2, 3.0, seed=78)
    for lv in (0, 1):
        for a, b in zip(out[lv], again[lv]):
            assert torch.equal(a, b)
    assert max_abs(out[1][0], other[1][0]) > 1e-5
    # stratification (helper.py:431-436): sample i lies in [lower_i, upper_i]
    lin = torch.linspace(0.0, 1.0, n0 + 1)
    edges = 0.2 * (1.0 - lin) + 3.0 * lin
    mids = 0.5 * (edges[1:] + edges[:-1])
    upper, lower = torch.cat([mids, edges[-1:]]), torch.cat([edges[:1], mids])
    t0 = ts[0].cpu()
    assert bool((t0 >= lower - 1e-6).all()) and bool((t0 <= upper + 1e-6).all()) and float(t0.std(dim=0).min()) > 1e-3
    assert ts[1].shape == (R, n0 + 1 + n1) and bool((ts[1][:, 1:] >= ts[1][:, :-1]).all())
    want = oracle.vanilla.render(synth.vanilla_state(0), rays_c, 0.2, 3.0, n_coarse=n0, n_fine=n1,
                                 samples=(ts[0].cpu(), ts[1].cpu()))
    for lv in (0, 1):
        for nm, a, b in zip(("rgb", "acc", "depth"), out[lv], want[lv]):
            assert max_abs(a, b) < 1e-4, (lv, nm)
    # the deterministic call under autograd takes the same operators and agrees with the fused kernels
    fused = net(rays, False, False, 0.2, 3.0)
    net.differentiable = True
    diff = net(rays, False, False, 0.2, 3.0)
    net.differentiable = None
    for lv in (0, 1):
        for nm, a, b in zip(("rgb", "acc", "depth"), diff[lv], fused[lv]):
            assert max_abs(a, b) < 1e-4, (lv, nm)


def test_vanilla_training_step_gradients_vs_fp64_autograd():
    """The reference's training_step (vanilla_nerf/model.py:281-300): loss = img2mse(coarse) + img2mse(fine) on 256 rays;
    gradients of all 48 parameter tensors against fp64 autograd of oracle.vanilla.render at the library's sample positions,
    under the noise-relative criterion of the NeO-360 module test: the library may miss the fp64 gradients by no more than
    1.5 x what the reference's own fp32 arithmetic misses them by (+ 2e-5)."""
    R, n0, n1 = 256, 16, 24
    sd = synth.vanilla_state(0)
    net = _vanilla(n0=n0, n1=n1)
    rays_c = cases.strided_rays(R)
    rays = {k: v.to(DEV) for k, v in rays_c.items()}
    target = synth.uniform(91, "vanilla_target", (R, 3), 0.0, 1.0)
    names = sorted(sd)
    with torch.enable_grad():
        for p in net.parameters():
            p.requires_grad_(True)
            p.grad = None
        out, ts = training.nerf_render_train(net, rays, True, False, 0.2, 3.0, seed=5, return_samples=True)
        assert out[0][0].requires_grad and out[1][0].requires_grad
        loss_g = ((out[0][0] - target.to(DEV)) ** 2).mean() + ((out[1][0] - target.to(DEV)) ** 2).mean()
        params = dict(net.named_parameters())
        g_g = torch.autograd.grad(loss_g, [params[k] for k in names])

        def oracle_grads(dtype):
            cv = lambda v: v.to(dtype)
            pp = {k: cv(v).clone().requires_grad_(True) for k, v in sd.items()}
            want = oracle.vanilla.render(pp, {k: cv(v) for k, v in rays_c.items()}, 0.2, 3.0, n_coarse=n0, n_fine=n1,
                                         samples=(cv(ts[0].cpu()), cv(ts[1].cpu())))
            loss = ((want[0][0] - cv(target)) ** 2).mean() + ((want[1][0] - cv(target)) ** 2).mean()
            return float(loss), torch.autograd.grad(loss, [pp[k] for k in names])

        loss_c, g_c = oracle_grads(torch.float64)
        _, g_r = oracle_grads(torch.float32)
    assert abs(float(loss_g) - loss_c) < 1e-5 * max(1.0, abs(loss_c))
    rel = lambda x, ref: (float(x.abs().max()) / (float(ref.abs().max()) + 1e-15), float(x.norm()) / (float(ref.norm()) + 1e-30))
    worst = 0.0
    for nm, a, b, r in zip(names, g_g, g_c, g_r):
        a = a.detach().cpu().double()
        lib, ref = rel(a - b, b), rel(r.double() - b, b)
        worst = max(worst, lib[1])
        assert lib[0] <= 1.5 * ref[0] + 2e-5 and lib[1] <= 1.5 * ref[1] + 2e-5, (nm, lib, ref)
    record_parity("train_vanilla_module_call", max_rel_l2_grad_err_vs_fp64=worst, rays=R, loss_abs_err=abs(float(loss_g) - loss_c))
    for p in net.parameters():
        p.requires_grad_(False)


def test_vanilla_noise_std_is_uniform_noise_on_the_raw_density():
    """model.py:194-195: raw_sigma + torch.rand_like(raw_sigma) * noise_std when randomized; the draws are streams 4 / 6 of the
    call's seed, so the oracle fed the same tables reproduces the outputs."""
    R, n0, n1, std, seed = 64, 16, 24, 0.75, 9
    rays_c = cases.strided_rays(R)
    rays = {k: v.to(DEV) for k, v in rays_c.items()}
    net = _vanilla(noise_std=std, n0=n0, n1=n1)
    out, ts = training.nerf_render_train(net, rays, True, False, 0.2, 3.0, seed=seed, return_samples=True)
    u = [training.rand_uniform(seed, 4 + 2 * lv, R, n).cpu() * std for lv, n in ((0, n0 + 1), (1, n0 + 1 + n1))]
    want = oracle.vanilla.render(synth.vanilla_state(0), rays_c, 0.2, 3.0, n_coarse=n0, n_fine=n1,
                                 samples=(ts[0].cpu(), ts[1].cpu()), sigma_noise=u)
    # level 0 only: level 1's positions depend on the noisy coarse weights (the same on both sides: they are ts[1])
    for lv in (0, 1):
        assert max_abs(out[lv][0], want[lv][0]) < 1e-4 and max_abs(out[lv][1], want[lv][1]) < 1e-4
    quiet = _vanilla(noise_std=0.0, n0=n0, n1=n1)
    assert max_abs(quiet(rays, True, False, 0.2, 3.0, seed=seed)[0][0], out[0][0]) > 1e-4
    # randomized=False ignores noise_std (the reference's `and randomized`)
    assert torch.equal(net(rays, False, False, 0.2, 3.0)[1][0], quiet(rays, False, False, 0.2, 3.0)[1][0])


def test_tp_density_noise_runs_on_the_operator_chain():
    sc = cases.small_scene()
    maps = [sc[k].to(DEV) for k in ("plane_xz", "plane_xy", "plane_yz", "latent")]
    gb = _batch(64)
    outs = {}
    for noise in (0.0, 1e-9, 0.5):
        net = models.NeRF_TP(num_coarse_samples=16, num_fine_samples=24, num_src_views=cases.NV, density_noise=noise).to(DEV)
        net.load_state_dict(synth.nerf_tp_state(0))
        net.set_scene(*maps, sc["image_wh"])
        outs[noise] = net(gb, True, False, 0.0, 0.0, out_depth=False, seed=31)      # no_grad (conftest), randomized, noise != 0 -> operators
    assert max_abs(outs[1e-9][0][0], outs[0.0][0][0]) < 1e-4                       # fused call vs operator chain, same samples
    assert max_abs(outs[0.5][0][0], outs[0.0][0][0]) > 1e-4
    assert torch.equal(outs[0.5][0][3], outs[0.0][0][3])                           # level-0 sample rows do not depend on the noise


def test_tp_training_call_chunked_draws_the_same_samples():
    """One seed, any chunk: the differentiable call slices ONE table of uniforms per stream (rows = rays of the call), which
    is what the fused neo_tp_render_train draws."""
    sc = cases.small_scene()
    maps = [sc[k].to(DEV) for k in ("plane_xz", "plane_xy", "plane_yz", "latent")]
    net = models.NeRF_TP(num_coarse_samples=16, num_fine_samples=24, num_src_views=cases.NV).to(DEV)
    net.load_state_dict(synth.nerf_tp_state(0))
    net.set_scene(*maps, sc["image_wh"])
    gb = _batch(96)
    fused = net(gb, True, False, 0.0, 0.0, out_depth=False, chunk=32, seed=1234)
    net.differentiable = True
    diff = net(gb, True, False, 0.0, 0.0, out_depth=False, chunk=32, seed=1234)
    net.differentiable = False
    again = net(gb, True, False, 0.0, 0.0, out_depth=False, chunk=32, seed=1234)
    assert torch.equal(diff[0][3], fused[0][3]) and torch.equal(diff[0][4], fused[0][4])      # level-0 rows of every chunk
    assert float((diff[1][3] - fused[1][3]).abs().median()) < 1e-6
    assert max_abs(diff[0][0], fused[0][0]) < 1e-4
    assert torch.equal(again[1][0], fused[1][0])


def test_differentiable_switch_is_explicit():
    net = _vanilla()
    rays = {k: v.to(DEV) for k, v in cases.strided_rays(8).items()}
    with torch.enable_grad():
        for p in net.parameters():
            p.requires_grad_(True)
        assert net._wants_grad()
        assert net(rays, False, False, 0.2, 3.0)[1][0].requires_grad
        net.differentiable = False                                   # forward-only caller outside no_grad: stays on the fused kernels
        assert not net(rays, False, False, 0.2, 3.0)[1][0].requires_grad
        for p in net.parameters():
            p.requires_grad_(False)
        net.differentiable = None
        assert not net._wants_grad()
    # frozen MLPs but scene tensors that want gradients: NeRF_TP's auto rule sees them
    sc = cases.small_scene()
    maps = [sc[k].to(DEV).clone() for k in ("plane_xz", "plane_xy", "plane_yz", "latent")]
    tp = models.NeRF_TP(num_coarse_samples=8, num_fine_samples=8, num_src_views=cases.NV).to(DEV)
    tp.load_state_dict(synth.nerf_tp_state(0))
    for p in tp.parameters():
        p.requires_grad_(False)
    with torch.enable_grad():
        maps[3].requires_grad_(True)
        tp.set_scene(*maps, sc["image_wh"])
        assert tp._wants_grad()
        out = tp(_batch(16), False, False, 0.0, 0.0, out_depth=False)
        assert out[1][0].requires_grad


# ---- PixelNeRF: exact fp32 evaluator -----------------------------------------------------------------------------------

def _pix(gain=1.0, precision=None):
    scene = cases.small_scene()
    net = models.PixelNeRF(num_src_views=cases.NV).to(DEV)
    net.precision = precision
    net.load_state_dict(synth.pixelnerf_state(0, density_gain=gain))
    net.set_scene(scene["latent"].to(DEV), scene["image_wh"])
    return net, scene


def test_pixelnerf_f32_stage_vs_oracle():
    net, scene = _pix(precision="f32")
    params = synth.pixelnerf_state(0)
    batch = cases.neo_batch(cases.strided_rays(128))
    gb = {k: v.to(DEV) for k, v in batch.items()}
    t = torch.sort(torch.rand(128, 97, generator=torch.Generator().manual_seed(3)) * 2.3 + 0.2, dim=-1).values
    for slot, prefix in ((0, "coarse_mlp."), (1, "fine_mlp.")):
        got = net.eval_mlp(slot, gb, t.to(DEV)).cpu()
        rgb, sigma = oracle.pixelnerf.region_eval(params, prefix, batch, scene, t)
        assert max_abs(got[..., :3], rgb) < 2e-5 and max_abs(got[..., 3:], sigma) < 2e-5, prefix
    # the chunk-dependent direction tiling (model_pixel.py:219-222)
    a, b = net.eval_mlp(0, gb, t.to(DEV), chunk=128).cpu(), net.eval_mlp(0, gb, t.to(DEV), chunk=64).cpu()
    assert max_abs(a[..., :3], b[..., :3]) > 1e-4 and max_abs(a[..., 3], b[..., 3]) == 0.0
    assert torch.equal(net.eval_mlp(0, gb, t.to(DEV)).cpu(), net.eval_mlp(0, gb, t.to(DEV)).cpu())
    # the two arithmetics agree to fp32 rounding
    split, _ = _pix()
    assert max_abs(split.eval_mlp(1, gb, t.to(DEV)), net.eval_mlp(1, gb, t.to(DEV))) < 5e-6


@pytest.mark.parametrize("tag,n_rays,chunk,gain,white", [("a", 300, 256, 1.0, False), ("sharp", 128, 128, 8.0, False),
                                                         ("white", 96, 96, 1.0, True)])
def test_pixelnerf_f32_end_to_end_vs_reference_fixture(golden, tag, n_rays, chunk, gain, white):
    g = golden("g7_pixelnerf")
    net, _ = _pix(gain, precision="f32")
    batch = {k: v.to(DEV) for k, v in cases.neo_batch(cases.strided_rays(n_rays)).items()}
    got = {k: [] for k in ("rgb0", "acc0", "depth0", "rgb1", "acc1", "depth1")}
    for i in range(0, n_rays, chunk):
        part = {k: (v[i:i + chunk] if k in PER_RAY else v) for k, v in batch.items()}
        res = net(part, False, white, 0.2, 2.5)
        for lv in (0, 1):
            got["rgb%d" % lv].append(res[lv][0]); got["acc%d" % lv].append(res[lv][1]); got["depth%d" % lv].append(res[lv][2])
    for k, v in got.items():
        assert max_abs(torch.cat(v).cpu(), g["%s_%s" % (k, tag)]) < 1e-4, (k, tag)


def test_pixelnerf_range_guard_retry_lands_on_the_exact_kernel():
    scene = cases.small_scene()
    big = scene["latent"].to(DEV) * 1.0e6
    batch = {k: v.to(DEV) for k, v in cases.neo_batch(cases.strided_rays(160)).items()}
    exact, _ = _pix(precision="f32")
    exact.set_scene(big, scene["image_wh"])
    want = render.render_rays_test(exact, batch, chunk=64, near=0.2, far=2.5)
    net, _ = _pix()
    net.set_scene(big, scene["image_wh"])
    with pytest.warns(RuntimeWarning, match="re-rendered on the exact fp32 kernels"):
        got = render.render_rays_test(net, batch, chunk=64, near=0.2, far=2.5)
    assert got["precision_used"] == "f32" and torch.equal(got["rgb"], want["rgb"]) and torch.equal(got["depth"], want["depth"])
    assert bool(torch.isfinite(got["rgb"]).all())


# ---- the range-guard latch ---------------------------------------------------------------------------------------------

def test_static_operand_trip_latches_until_weights_or_scene_change():
    """A feature map beyond the fp16 range trips the guard on EVERY frame: after the first (split attempt + exact retry) the
    module goes straight to the exact kernels - one render per frame - until set_scene / load_state_dict."""
    sc = dict(cases.small_scene())
    sc["latent"] = sc["latent"] * 1.0e7          # beyond the fp16 range even after the projection through the first layer
    batch = _batch(128)
    net = _tp_net(sc)
    ctx = net._context(torch.device(DEV))
    assert net._range_latch is None
    with pytest.warns(RuntimeWarning):
        first = render.render_rays_test(net, batch, chunk=64)
    assert first["precision_used"] == "f32" and net._range_latch is not None and net.last_precision_used == "f32"

    def evaluator_launches(fn):
        ctx.set_timing(True)
        out = fn()
        torch.cuda.synchronize()
        n = len(ctx.read_spans())
        ctx.set_timing(False)
        return out, n

    with warnings.catch_warnings():
        warnings.simplefilter("error")
        second, n = evaluator_launches(lambda: render.render_rays_test(net, batch, chunk=64))
    assert second["precision_used"] == "f32" and torch.equal(second["rgb"], first["rgb"])
    assert n == 4, n                                   # ONE frame's four evaluator launches, not a failed attempt + a retry (8)
    assert net.precision is None
    # a healthy scene: the latch is dropped, the split kernels run again
    ok = cases.small_scene()
    net.set_scene(ok["plane_xz"].to(DEV), ok["plane_xy"].to(DEV), ok["plane_yz"].to(DEV), ok["latent"].to(DEV), ok["image_wh"])
    fine = render.render_rays_test(net, batch, chunk=64)
    assert "precision_used" not in fine and net._range_latch is None and net.last_precision_used == "f16x3"
    # the sharded render reports the arithmetic it used
    net2 = _tp_net(sc)
    info = {}
    with pytest.warns(RuntimeWarning):
        render.render_frame_sharded(net2, batch, 1, 0, chunk=64, info=info)
    assert info["precision_used"] == "f32"
    info = {}
    render.render_frame_sharded(_tp_net(ok), batch, 1, 0, chunk=64, info=info)
    assert info["precision_used"] == "f16x3"


# ---- bench.py: both launch forms, scene set-up from events ----------------------------------------------------------------

def _bench(cmd_prefix, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    r = subprocess.run(cmd_prefix + [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--cpu-rays", "0",
                                    "--others", "0", "--exact-f32", "0"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rows = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(rows) == 1
    return json.loads(rows[0])


def test_bench_plain_form_scene_setup_from_events():
    out = _bench([sys.executable])
    assert out["n_gpus"] == 1 and out["unit"] == "rays/s" and out["value"] > 1e5
    assert out["config"]["launch"] == "single process"
    roof = out["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and "limiter" in roof and 0.0 < roof["frac"] < 1.0
    ss = out["scene_setup"]
    runs = ss["runs_ms"]
    assert len(runs) == 7 and out["scene_setup_ms"] == ss["total_ms"]
    # The number of record is the second smallest run.  No bound on its size or on the runs' agreement here: the event window holds
    # ~150 enqueues and 72 small H2D copies issued by the host thread, and the box decides how long those take - the builder's boxes
    # gave 7.2-7.7 ms on most, 93 / 95 / 7.6 / 7.7 / 15 / 174 ms within one call on a loaded one, and a steady 94-100 ms on another
    # (GPU work in the window: ~5 ms in rocprofv3).  A test that asserted either would fail the whole suite on such a box.
    assert ss["total_ms"] == sorted(runs)[1] and 0.5 < ss["total_ms"] < 5000.0, ss


# ---- the training call's fused point / activation operators (round 5) ------------------------------------------------------

def test_train_points_match_the_oracle():
    """neo_tp_train_points = neo360/helper.py:24-75 (+ :401-451 outside the sphere) + util.py:52-70 + pos_enc, per region."""
    R, N = 48, 33
    rays_c = cases.strided_rays(R)
    batch = cases.neo_batch(rays_c)
    rays = {k: v.to(DEV) for k, v in batch.items()}
    far, _ = ops.intersect_sphere(rays["rays_o"], rays["rays_d"])
    fg_t, bg_s = training.sample_level0(far, N - 1)
    far_c, _ = oracle.rays.sphere_exit_depth(batch["rays_o"], batch["rays_d"])
    poses = batch["src_poses"]
    # inside: o + t d, camera-frame 63-d encodings per view
    look, x = training.train_points(None, 0, rays["rays_o"], rays["rays_d"], fg_t, None, rays["src_poses"])
    pts = oracle.sampling.points_on_rays(fg_t.cpu(), batch["rays_o"], batch["rays_d"]).reshape(-1, 3)
    assert max_abs(look, pts) < 1e-6
    want = oracle.encoding.pos_enc(oracle.gather.world_to_camera(pts, poses), 0, 10)
    assert x.shape == want.shape == (cases.NV, R * N, 63) and max_abs(x, want) < 1e-4 and float((x.cpu() - want).abs().median()) < 1e-6   # octave 9 amplifies the last bit of a camera-frame coordinate 512 x
    # outside: linear lookup point, inverted-sphere point in the camera frame + the inverse radius, 84-d encodings
    look, x = training.train_points(None, 1, rays["rays_o"], rays["rays_d"], bg_s, far, rays["src_poses"])
    s = bg_s.cpu()
    lin = batch["rays_o"][:, None, :] + (far_c.reshape(-1, 1) * (1.0 - s) + 3.0 * s)[..., None] * batch["rays_d"][:, None, :]
    assert max_abs(look, lin.reshape(-1, 3)) < 2e-6
    p4 = oracle.sampling.inverted_sphere_points(batch["rays_o"], batch["rays_d"], s)
    cam = torch.cat((oracle.gather.world_to_camera(p4[..., :3].reshape(-1, 3), poses),
                     p4[..., 3].reshape(1, -1, 1).expand(cases.NV, -1, -1)), dim=-1)
    want = oracle.encoding.pos_enc(cam, 0, 10)
    assert x.shape == want.shape == (cases.NV, R * N, 84) and max_abs(x, want) < 2e-4 and float((x.cpu() - want).abs().median()) < 1e-6


def test_activate_op_matches_autograd():
    P = 4096
    g = torch.Generator().manual_seed(5)
    raw_rgb, raw_sigma = torch.randn(P, 3, generator=g) * 3.0, torch.randn(P, 1, generator=g) * 12.0     # beyond the softplus threshold too
    noise, scale = torch.rand(P, generator=g), 0.7
    up = torch.randn(P, 4, generator=g)
    with torch.enable_grad():
        a, b = raw_rgb.clone().double().requires_grad_(True), raw_sigma.clone().double().requires_grad_(True)
        want = torch.cat([torch.sigmoid(a) * 1.002 - 0.001, torch.nn.functional.softplus(b + (noise.double() * scale)[:, None] - 1.0)], dim=-1)
        (want * up.double()).sum().backward()
        ga, gb = raw_rgb.to(DEV).requires_grad_(True), raw_sigma.to(DEV).requires_grad_(True)
        got = training.activate(ga, gb, noise.to(DEV), scale)
        (got * up.to(DEV)).sum().backward()
    assert got.shape == (P, 4) and max_abs(got, want.detach().float()) < 2e-6
    assert max_abs(ga.grad, a.grad.float()) < 2e-6 and max_abs(gb.grad, b.grad.float()) < 2e-6
    plain = training.activate(raw_rgb.to(DEV), raw_sigma.to(DEV))
    assert max_abs(plain[:, 3], torch.nn.functional.softplus(raw_sigma[:, 0] - 1.0)) < 2e-6


def test_projected_space_training_equals_the_per_row_path():
    """Round 5: the training analogue of the pre-projection - the latent goes through [W0_loc | W3_loc] once per call in texel
    space (a library GEMM under autograd), 256 channels are gathered and handed to the MLP as `pre`.  A reassociation: same
    forward values, same gradients for EVERY parameter (the local weight columns included: their gradient now comes out of the
    texel-space GEMM's backward), the three tri-planes and the latent, as the per-row path (which the fp64-autograd test of
    tests/test_gpu_training.py pins to the oracle)."""
    sc = cases.small_scene()
    R = 96
    gb = _batch(R)
    target = synth.uniform(17, "proj_target", (R, 3), 0.0, 1.0).to(DEV)
    results = {}
    for projected in (False, True):
        net = models.NeRF_TP(num_coarse_samples=16, num_fine_samples=24, num_src_views=cases.NV).to(DEV)
        net.load_state_dict(synth.nerf_tp_state(0))
        net.train_projected = projected
        maps = [sc[k].to(DEV).clone().requires_grad_(True) for k in ("plane_xz", "plane_xy", "plane_yz", "latent")]
        with torch.enable_grad():
            net.set_scene(*maps, sc["image_wh"])
            for p in net.parameters():
                p.requires_grad_(True)
            out = net(gb, True, False, 0.0, 0.0, out_depth=False, seed=99)
            loss = sum(((lv[0] - target) ** 2).mean() for lv in out) + 0.01 * training.eff_distloss(out[1][1], out[1][3], 1.0 / 41)
            names = sorted(n for n, _ in net.named_parameters())
            params = dict(net.named_parameters())
            grads = torch.autograd.grad(loss, [params[n] for n in names] + maps)
        results[projected] = (float(loss), [o.detach() for lv in out for o in lv], names, [g.detach() for g in grads])
    (l0, o0, names, g0), (l1, o1, _, g1) = results[False], results[True]
    assert abs(l0 - l1) < 1e-6 * max(1.0, abs(l0))
    for a, b in zip(o0, o1):
        assert float((a - b).abs().median()) < 1e-6 and max_abs(a, b) < 1e-3         # fine-level rows may flip a resampled position
    worst = 0.0
    for nm, a, b in zip(names + ["plane_xz", "plane_xy", "plane_yz", "latent"], g0, g1):
        rel = float((a - b).norm()) / (float(a.norm()) + 1e-20)
        worst = max(worst, rel)
        assert rel < 2e-3, (nm, rel)
        assert float(b.abs().max()) > 0.0, nm                                            # nothing lost: every tensor receives a gradient
    # the local columns of the first layer really are trained through the texel-space path
    i0 = names.index("fg_fine_mlp.pts_linears.0.weight")
    assert float(g1[i0][:, 63:63 + 512].abs().max()) > 0.0
    record_parity("train_projected_vs_per_row", worst_rel_l2_grad_diff=worst, loss_diff=abs(l0 - l1), rays=R)


@pytest.mark.parametrize("K,M,N", [(5000, 256, 512), (1031, 3, 64), (70001, 128, 63), (40000, 64, 27), (2500, 1, 128), (33, 300, 130)])
def test_linear_weight_grad_matches_fp64(K, M, N):
    """neo_linear_weight_grad (k_dw + k_dw_reduce: 128 x 128 tiles over K slices, partial tiles summed by a second kernel): dW = gy^T x
    and db = column sums of gy against fp64, on full tiles, edge tiles (M = 1, 3, 300; N = 27, 63, 130) and a K that is not a
    multiple of the K step; gradients span orders of magnitude (rows scaled by 10^U(-4, 0))."""
    g = torch.Generator(device=DEV).manual_seed(K + M + N)
    scale = 10.0 ** (-4.0 * torch.rand(K, 1, device=DEV, generator=g))
    gy = torch.randn(K, M, device=DEV, generator=g) * scale
    x = torch.randn(K, N, device=DEV, generator=g)
    gw, gb = training.weight_grad(gy, x, bias=True)
    ref_w = gy.double().t() @ x.double()
    ref_b = gy.double().sum(0)
    tol = 2e-6 * float(ref_w.abs().max())
    assert float((gw.double() - ref_w).abs().max()) <= tol
    assert float((gb.double() - ref_b).abs().max()) <= 2e-6 * max(float(ref_b.abs().max()), float(gy.abs().sum(0).max()) * 1e-3)
    # without the bias gradient, and accumulation onto the caller's buffer semantics (a fresh zeroed buffer per call)
    gw2 = training.weight_grad(gy, x)
    assert float((gw2.double() - ref_w).abs().max()) <= tol


def test_project_latent_backward_matches_autograd_of_matmul():
    """training.project_latent's custom backward (library GEMM for the latent, neo_linear_weight_grad for the two weight blocks)
    equals autograd of the plain matmul."""
    mlp = models.NeRFPPMLP(0, 10, 4, input_ch=3, num_src_views=3).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(3)
    lat = (torch.randn(6000, 512, device=DEV, generator=g) * 0.3).requires_grad_(True)
    up = torch.randn(6000, 256, device=DEV, generator=g)
    with torch.enable_grad():
        out = training.project_latent(mlp, lat)
        (out * up).sum().backward()
        got = [lat.grad.clone(), mlp.pts_linears[0].weight.grad.clone(), mlp.pts_linears[3].weight.grad.clone()]
        lat.grad = None
        mlp.zero_grad()
        w0, w3 = mlp.pts_linears[0].weight, mlp.pts_linears[3].weight
        ref = lat @ torch.cat([w0[:, 63:575], w3[:, 191:703]], 0).t()
        assert max_abs(out, ref) <= 1e-5
        (ref * up).sum().backward()
    for a, b in zip(got, [lat.grad, w0.grad, w3.grad]):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
