"""Timing of the scene encoder's pillar stage (csrc/pillar.hip) at the reference grid (64^3 cells, 3 source views):
library call vs the CPU oracle on a slice.  env: REPS, GRID (e.g. 64x64x64)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neo360_amd import encoder, synth
torch.set_grad_enabled(False)
dev = "cuda"
grid = tuple(int(x) for x in os.environ.get("GRID", "64x64x64").split("x"))
reps = int(os.environ.get("REPS", 10))
g = torch.Generator(device=dev); g.manual_seed(0)
NV = 3
latent = torch.randn(NV, 512, 60, 80, device=dev, generator=g) * 0.3
poses, focal, centre = synth.source_views(NV, 640, 480)
enc = encoder.GridEncoder(grid_size=grid).to(dev)
enc.load_state_dict(synth.pillar_state(0), strict=False)
args = (latent, poses.to(dev), focal.to(dev), centre.to(dev), (640.0, 480.0))
for _ in range(2): out = enc.floorplans(*args)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): out = enc.floorplans(*args)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
cells = grid[0] * grid[1] * grid[2] * NV
macs = 518 * 512 + 2 * 512 * 512 + 3 * (513 * 512 + 512)          # encoder_tp_fusion_conv.py:263-279, :364-373
print("pillar stage grid %s x %d views: %.2f ms per call, %.1f algorithmic TFLOP/s (%.1f %% of 833), %.0f M cell-views/s, checksum %.6f"
      % ("x".join(map(str, grid)), NV, dt * 1e3, cells * macs * 2 / dt / 1e12, cells * macs * 2 / dt / 1e12 / 8.333, cells / dt / 1e6,
         float(sum(o.double().sum() for o in out))))
