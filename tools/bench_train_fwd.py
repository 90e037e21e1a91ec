"""Forward-only throughput of the vanilla NeRFMLP training op on 790 k rows (used with the NEO_SGEMM_ABLATE variants of
train_mlp.hip: profiles/r03_train_mlp_bench.log)."""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neo360_amd import models, synth, training
dev = "cuda"
vm = models.NeRFMLP().to(dev)
sd = synth.vanilla_state(0)
vm.load_state_dict({k[len("fine_mlp."):]: v for k, v in sd.items() if k.startswith("fine_mlp.")})
x, d = torch.randn(4096, 193, 63, device=dev), torch.randn(4096, 27, device=dev)
fl = 2.0 * 4096 * 193 * (63 * 256 + 6 * 256 * 256 + 319 * 256 + 256 + 256 * 256 + 283 * 128 + 128 * 3)
with torch.no_grad():
    training.nerf_mlp(vm, x, d); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): training.nerf_mlp(vm, x, d)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
print("vanilla forward 790k rows: %.2f ms  %.1f TFLOP/s" % (t * 1e3, fl / t / 1e12))
