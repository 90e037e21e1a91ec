"""Throughput of the training-side NeRFPPMLP (neo_tp_mlp_train_forward / _backward) on one reference-sized chunk level:
1024 rays x 385 fine samples x 3 views = 1.18 M rows (env P = points, NV).  Forward 255,424 MAC per row (+ per-point head),
backward 2x that."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neo360_amd import models, synth, training
dev = "cuda"
NV, P = int(os.environ.get("NV", 3)), int(os.environ.get("P", 1024 * 385))
mlp = models.NeRFPPMLP(0, 10, 4, input_ch=3, num_src_views=NV).to(dev)
sd = synth.nerf_tp_state(0)
mlp.load_state_dict({k[len("fg_fine_mlp."):]: v for k, v in sd.items() if k.startswith("fg_fine_mlp.")})
g = torch.Generator(device=dev).manual_seed(0)
x_enc = torch.randn(NV, P, 63, device=dev, generator=g)
cond = torch.randn(NV * P, 27, device=dev, generator=g)
world = (torch.randn(NV * P, 128, device=dev, generator=g) * 0.3).requires_grad_(True)
local = (torch.randn(NV * P, 512, device=dev, generator=g) * 0.3).requires_grad_(True)
macs_row = 703 * 128 + 2 * 128 * 128 + 831 * 128 + 128 * 128 + 155 * 64
macs_pt = 128 + 64 * 64 + 64 * 3
fwd_flop = 2.0 * (NV * P * macs_row + P * macs_pt)
def step():
    rgb, sig = training.nerfpp_mlp(mlp, x_enc, cond, world, local, NV)
    (rgb.sum() + sig.sum()).backward()
with torch.enable_grad():
    step(); step(); step(); torch.cuda.synchronize()      # the caching allocator settles after the second backward
    t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(3): training.nerfpp_mlp(mlp, x_enc, cond, world, local, NV)
    torch.cuda.synchronize(); tf = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    for _ in range(3): step()
    torch.cuda.synchronize(); ts = (time.perf_counter() - t0) / 3
print("NeRFPPMLP training op, %d rows (%d points x %d views): forward %.1f ms (%.1f TFLOP/s), forward + backward %.1f ms (%.1f TFLOP/s of 3 x forward flops; exact fp32 MFMA peak 157.3)"
      % (NV * P, P, NV, tf * 1e3, fwd_flop / tf / 1e12, ts * 1e3, 3 * fwd_flop / ts / 1e12))

# ---- the vanilla NeRFMLP (neo_vanilla_mlp_train_forward / _backward): 1024 rays x 193 fine samples, 593,408 MAC per row ----
B, N = int(os.environ.get("VB", 1024)), int(os.environ.get("VN", 193))
vmlp = models.NeRFMLP().to(dev)
vsd = synth.vanilla_state(0)
vmlp.load_state_dict({k[len("fine_mlp."):]: v for k, v in vsd.items() if k.startswith("fine_mlp.")})
vx = torch.randn(B, N, 63, device=dev, generator=g).requires_grad_(True)
vd = torch.randn(B, 27, device=dev, generator=g)
vflop = 2.0 * B * N * (63 * 256 + 6 * 256 * 256 + 319 * 256 + 256 + 256 * 256 + 283 * 128 + 128 * 3)
def vstep():
    rgb, sig = training.nerf_mlp(vmlp, vx, vd)
    (rgb.sum() + sig.sum()).backward()
with torch.enable_grad():
    vstep(); vstep(); vstep(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(3): training.nerf_mlp(vmlp, vx, vd)
    torch.cuda.synchronize(); tf = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    for _ in range(3): vstep()
    torch.cuda.synchronize(); ts = (time.perf_counter() - t0) / 3
print("NeRFMLP (vanilla) training op, %d rows: forward %.1f ms (%.1f TFLOP/s), forward + backward %.1f ms (%.1f TFLOP/s of 3 x forward flops)"
      % (B * N, tf * 1e3, vflop / tf / 1e12, ts * 1e3, 3 * vflop / ts / 1e12))
