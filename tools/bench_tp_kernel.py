"""Micro-benchmark of the NeO-360 point-evaluator kernel alone (env: PREC=f16x3|f32, R, N, SLOT or SLOTS=1,3,0,2, REPS,
PP=0|1|2 pre-projection mode).  SLOTS runs several slots in one process (N = 385 for the fine slots 1 / 3, 129 for 0 / 2) and
prints socket power / shader clock sampled during each timed loop (neo360_amd.telemetry).
Same synthetic scene as bench.py --workload neo360 (3 source views, 240x320 latent, 120x160 planes)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neo360_amd import models, synth, ops
torch.set_grad_enabled(False)
dev = torch.device("cuda")
PREC = os.environ.get("PREC", "f16x3")
R, N = int(os.environ.get("R", 8192)), int(os.environ.get("N", 385))
SLOT, REPS = int(os.environ.get("SLOT", 1)), int(os.environ.get("REPS", 3))
NV = 3
net = models.NeRF_TP(num_src_views=NV).to(dev)
net.precision = PREC
if os.environ.get("PP") is not None:
    net.preproject = {"0": False, "1": True, "2": 2, "3": 3}[os.environ["PP"]]
net.poll_flags = os.environ.get("POLL", "1") != "0"       # POLL=0: timing ablations produce garbage operands
SCALE = float(os.environ.get("SCALE", 1.0))      # 0: all-zero weights and features (power / clock envelope experiments)
SCALE_W = float(os.environ.get("SCALE_W", SCALE))   # weights only / features only: which operands carry the power
SCALE_F = float(os.environ.get("SCALE_F", SCALE))
net.load_state_dict({k: v * SCALE_W for k, v in synth.nerf_tp_state(0).items()})
H, W, focal = 480, 640, 512.0
g = torch.Generator(device=dev); g.manual_seed(0)
planes = [torch.randn(NV, 128, 120, 160, device=dev, generator=g) * (0.1 * SCALE_F) for _ in range(3)]
latent = torch.randn(NV, 512, 240, 320, device=dev, generator=g) * (0.1 * SCALE_F)
net.set_scene(planes[0], planes[1], planes[2], latent, (float(W), float(H)))
c2w = synth.look_at_origin(40.0)
ro, vd, rd, _ = ops.get_ray_directions_and_rays(H, W, focal, c2w)
sel = torch.randperm(H * W, device=dev, generator=torch.Generator(device=dev).manual_seed(1))[:R]
poses, sfocal, centre = synth.source_views(NV, W, H)
rays = {"rays_o": ro[sel].contiguous(), "rays_d": rd[sel].contiguous(), "viewdirs": vd[sel].contiguous(),
        "src_poses": poses.to(dev), "src_focal": sfocal.to(dev), "src_c": centre.to(dev)}
far, _ = ops.intersect_sphere(rays["rays_o"], rays["rays_d"])
_eval = net.eval_mlp
def eval_mlp(*a, **k):
    try:
        return _eval(*a, **k)
    except Exception as e:                      # POLL=0: timing ablations feed garbage operands and trip the range guard
        if os.environ.get("POLL", "1") != "0" or "fp16 range" not in str(e):
            raise
        return torch.zeros(1, device=dev)
net.eval_mlp = eval_mlp
from neo360_amd import telemetry
SLOTS = [int(x) for x in os.environ["SLOTS"].split(",")] if os.environ.get("SLOTS") else [SLOT]
for SLOT in SLOTS:
    if os.environ.get("SLOTS"):
        N = 385 if SLOT in (1, 3) else 129
    if SLOT < 2:
        t = torch.linspace(0.02, 0.98, N, device=dev)[None, :] * far.reshape(-1, 1)
    else:
        t = torch.linspace(0.98, 0.02, N, device=dev)[None, :].expand(R, N).contiguous()
    for _ in range(2):
        net.eval_mlp(SLOT, rays, t, far=far)
    torch.cuda.synchronize()
    with telemetry.Sampler(0, period_s=0.02) as tel:
        t0 = time.perf_counter()
        for _ in range(REPS):
            out = net.eval_mlp(SLOT, rays, t, far=far)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / REPS
    ts = tel.summary()
    pe = 63 if SLOT < 2 else 84
    macs = NV * ((pe + 640) * 128 + 2 * 128 * 128 + (pe + 640 + 128) * 128 + 128 * 128 + 155 * 64) + 128 + 64 * 64 + 64 * 3
    print("%s %s pp=%s slot %d R=%d N=%d  %.3f ms  %.1f algorithmic TFLOP/s  checksum %.6f  sclk %s MHz  power %s W (%s samples)  energy/launch %s J (accumulator)  %s J (power x time)" % (
        os.environ.get("TAG", ""), PREC, int(net.preproject), SLOT, R, N, dt * 1e3, R * N * macs * 2 / dt / 1e12, float(out.double().sum()),
        ("%.0f" % ts["sclk_mhz_mean"]) if ts.get("sclk_mhz_mean") else "?", ("%.0f" % ts["power_w_mean"]) if ts.get("power_w_mean") else "?",
        ts.get("telemetry_samples", 0), ("%.3f" % (ts["energy_j"] / REPS)) if ts.get("energy_j") else "?",
        ("%.3f" % (ts["power_w_mean"] * dt)) if ts.get("power_w_mean") else "?"), flush=True)
if os.environ.get("TRACE"):
    # variant built with -DNEO_TP_TRACE=1: per-phase s_memtime sums of wave 0 of every workgroup (k_tp_mlp_hp; TRACE=hpp: k_tp_mlp_hpp)
    import ctypes
    from neo360_amd import _lib
    lib = _lib.load()
    buf = (ctypes.c_ulonglong * 16)()
    hpp = os.environ["TRACE"] == "hpp"
    f32 = os.environ["TRACE"] == "f32"          # library built -DNEO_TP32_TRACE=1: the exact-fp32 evaluator k_tp_mlp (PREC=f32)
    read = lib.neo_debug_tp32_trace if f32 else lib.neo_debug_tpp_trace if hpp else lib.neo_debug_tp_trace
    read(buf, 1)                 # reset (drops warm-up + timed launches above)
    net.eval_mlp(SLOT, rays, t, far=far)
    read(buf, 0)
    names = (["setup", "descriptors", "projected maps: gather + add", "streamed stages (pos_enc)", "L0 epi + L1 + L2", "L3 + view sums", "tail"] if f32 else
             ["setup", "descriptors + pos_enc", "work list + prologue", "gather + pos_enc k-steps + adds", "L0 epi + L1..L3", "tail"] if hpp else
             ["setup", "descriptors", "G gather+consume", "planes (+X ks0-3)", "X ks4.. + pos_enc", "L0 epi + L1..L3", "tail"])
    n = max(int(buf[7]), 1)
    tot = sum(int(buf[k]) for k in range(len(names)))
    print("%s phase trace slot %d: %d workgroups, %.0f cycles (s_memtime ticks, 100 MHz) per tile" % (os.environ.get("TAG", ""), SLOT, n, tot / n))
    for k, nm in enumerate(names):
        print("   %-32s %10.1f per tile  (%5.1f %%)%s" % (nm, int(buf[k]) / n, 100.0 * int(buf[k]) / max(tot, 1),
                                                       "   [per view: %.1f]" % (int(buf[k]) / n / NV) if 1 <= k <= len(names) - 2 else ""))
