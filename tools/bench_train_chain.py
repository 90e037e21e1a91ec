"""The projected-space NeRFPPMLP training op alone (neo_tp_mlp_train_forward_pre / _backward_pre; csrc/train_chain.h when
neo_train_chain_mode = 1): forward and forward + backward times on one fine level of the bench's training step, 500 rays x 385 samples
x 3 views = 577,500 rows (env P, NV, CH).  Forward 248 KFLOP per row, backward chain 213 KFLOP per row + the weight-gradient GEMMs."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neo360_amd import _lib, models, synth, training
dev = "cuda"
NV, P, CH = int(os.environ.get("NV", 3)), int(os.environ.get("P", 500 * 385)), int(os.environ.get("CH", 3))
mlp = models.NeRFPPMLP(0, 10, 4, input_ch=CH, num_src_views=NV).to(dev)
g = torch.Generator(device=dev).manual_seed(0)
with torch.no_grad():
    for p in mlp.parameters():
        p.copy_(torch.randn(p.shape, device=dev, generator=g) * (0.1 if p.dim() == 2 else 0.02))
x_enc = torch.randn(NV, P, 21 * CH, device=dev, generator=g)
cond = torch.randn(NV * P, 27, device=dev, generator=g)
world = (torch.randn(NV * P, 128, device=dev, generator=g) * 0.3).requires_grad_(True)
pre = (torch.randn(NV * P, 256, device=dev, generator=g) * 0.3).requires_grad_(True)
pe = 21 * CH
# executed flops: mode 1 runs the bottleneck and view layer 0 on the view means (P rows), mode 0 per row (NV P rows)
row_mac, head_mac = 128 * (pe + 128) + 2 * 128 * 128 + 128 * (128 + pe + 128), 128 * 128 + 64 * 155
mode = _lib.load().neo_train_chain_mode(-1)
fwd_flop = 2.0 * (NV * P * row_mac + (P if mode else NV * P) * head_mac + P * (128 + 64 * 64 + 3 * 64))
def step():
    rgb, sig = training.nerfpp_mlp_projected(mlp, x_enc, cond, world, pre, NV)
    (rgb.sum() + sig.sum()).backward()
def timed(fn, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
with torch.enable_grad():
    for _ in range(3): step()
    def fwd():
        with torch.no_grad(): training.nerfpp_mlp_projected(mlp, x_enc, cond, world, pre, NV)
    tf = timed(fwd); ts = timed(step)
print("%s chain mode %d: %d rows: forward %.3f ms (%.1f TFLOP/s of 157.3), forward + backward %.3f ms"
      % (os.environ.get("TAG", ""), _lib.load().neo_train_chain_mode(-1), NV * P, tf * 1e3, fwd_flop / tf / 1e12, ts * 1e3))
