import os, sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import oracle, cases
from neo360_amd import synth
torch.set_grad_enabled(False)
state = synth.vanilla_state(0)
rays = cases.strided_rays(1024)
for th in (8, 16, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1): break
    torch.set_num_threads(th)
    oracle.vanilla.render(state, {k: v[:64] for k, v in rays.items()}, 0.2, 3.0)
    t0 = time.perf_counter(); oracle.vanilla.render(state, rays, 0.2, 3.0); dt = time.perf_counter() - t0
    print("threads", th, "rays/s %.1f" % (1024 / dt), flush=True)
