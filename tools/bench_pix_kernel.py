"""Micro-benchmark of the PixelNeRF point-evaluator kernel alone (env: R, N, SLOT, REPS)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neo360_amd import models, synth, ops
torch.set_grad_enabled(False)
dev = torch.device("cuda")
R, N = int(os.environ.get("R", 16384)), int(os.environ.get("N", 129))
SLOT, REPS = int(os.environ.get("SLOT", 1)), int(os.environ.get("REPS", 5))
NV, H, W = 3, 480, 640
net = models.PixelNeRF(num_src_views=NV).to(dev)
net.load_state_dict(synth.pixelnerf_state(0))
g = torch.Generator(device=dev); g.manual_seed(0)
net.set_scene(torch.randn(NV, 512, 240, 320, device=dev, generator=g) * 0.1, (float(W), float(H)))
ro, vd, rd, _ = ops.get_ray_directions_and_rays(H, W, 512.0, synth.look_at_origin(40.0))
poses, sfocal, centre = synth.source_views(NV, W, H)
rays = {"rays_o": ro[:R].contiguous(), "rays_d": rd[:R].contiguous(), "viewdirs": vd[:R].contiguous(),
        "src_poses": poses.to(dev), "src_focal": sfocal.to(dev), "src_c": centre.to(dev)}
t = torch.linspace(0.2, 3.0, N, device=dev)[None, :].expand(R, N).contiguous()
net.eval_mlp(SLOT, rays, t)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(REPS):
    out = net.eval_mlp(SLOT, rays, t)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / REPS
macs = NV * 158976 + 16896
print("%s pix slot %d R=%d N=%d  %.2f ms  %.1f algorithmic TFLOP/s  checksum %.6f" % (
    os.environ.get("TAG", ""), SLOT, R, N, dt * 1e3, R * N * macs * 2 / dt / 1e12, float(out.double().sum())))
