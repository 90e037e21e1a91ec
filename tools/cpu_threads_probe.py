"""Thread-count sweep of the CPU oracle on the GPU box's host, per workload (WORKLOAD=neo360|vanilla).  NeO-360: a
256-ray chunk of the full-size C3 configuration (128 + 256 samples, 3 views, full-size feature maps) per thread count;
bench.py's CPU_THREADS is the fastest setting of this sweep (profiles/r03_cpu_threads_neo360.log)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import oracle, cases
from neo360_amd import synth
torch.set_grad_enabled(False)
wl = os.environ.get("WORKLOAD", "neo360")
print("host: %d logical CPUs; workload %s" % (os.cpu_count() or 1, wl), flush=True)
if wl == "vanilla":
    state = synth.vanilla_state(0)
    rays = cases.strided_rays(1024)
    run = lambda n: oracle.vanilla.render(state, {k: v[:n] for k, v in rays.items()}, 0.2, 3.0)
    n_small, n_big = 64, 1024
else:
    state = synth.nerf_tp_state(0)
    scene = cases.full_scene()
    batch = cases.full_batch(256)
    per_ray = ("rays_o", "rays_d", "viewdirs")
    run = lambda n: oracle.neo360.render(state, {k: (v[:n] if k in per_ray else v) for k, v in batch.items()}, scene, 128, 256)
    n_small, n_big = 16, 256
for th in (8, 16, 32, 48, 64, 96, 128, 256):
    if th > (os.cpu_count() or 1):
        break
    torch.set_num_threads(th)
    run(n_small)
    t0 = time.perf_counter(); run(n_big); dt = time.perf_counter() - t0
    print("threads %3d  %7.2f rays/s  (%d rays in %.1f s)" % (th, n_big / dt, n_big, dt), flush=True)
