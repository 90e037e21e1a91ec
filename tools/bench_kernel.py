"""Micro-benchmark of the fused vanilla MLP kernel alone (variants via env vars)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from neo360_amd import models, synth, ops
torch.set_grad_enabled(False)
dev = torch.device("cuda")
# SCALE multiplies every parameter: 0 = all-zero operands (same instruction stream, no data toggling in the matrix pipes)
state = {k: v * float(os.environ.get("SCALE", 1.0)) for k, v in synth.vanilla_state(0).items()}
net = models.NeRF().to(dev); net.load_state_dict(state); net.precision = os.environ.get('PREC', 'f32')
R, N = int(os.environ.get("R", 65536)), int(os.environ.get("N", 193))
ro, vd, rd, _ = ops.get_ray_directions_and_rays(480, 640, 512.0, synth.look_at_origin(40.0))
ro, vd = ro[:R].contiguous(), vd[:R].contiguous()
t = torch.sort(torch.rand(R, N, device=dev) * 2.8 + 0.2, dim=-1).values
for _ in range(2): net.eval_mlp(1, ro, vd, t)
torch.cuda.synchronize()
reps = int(os.environ.get('REPS', 5))
t0 = time.perf_counter()
for _ in range(reps): out = net.eval_mlp(1, ro, vd, t)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print("%s R=%d N=%d  %.2f ms  %.1f TFLOP/s (%.1f%% of 157.3)  checksum %.6f" % (
    os.environ.get("TAG", ""), R, N, dt * 1e3, R * N * 1186816 / dt / 1e12, R * N * 1186816 / dt / 1e12 / 1.573, float(out.double().sum())))
