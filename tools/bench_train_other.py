"""Training-step times of the three renderers whose step is not the bench.py line: forward (randomized) + a training_step-shaped loss +
backward + Adam, reference batch size 1024 rays (opt.py:188).  PixelNeRF: 3 views, 64 + 64 samples, 240 x 320 latent; Mip-NeRF 360:
64 / 64 / 32 intervals; vanilla: 64 + 128."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases
from neo360_amd import models, synth
dev = "cuda"
R = int(os.environ.get("RAYS", 1024))
torch.set_grad_enabled(True)


def timed(name, step, n=6):
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("%-12s %4d rays: %.1f ms per step (%.0f rays/s)" % (name, R, dt * 1e3, R / dt))


rays = {k: v.to(dev) for k, v in cases.strided_rays(R).items()}
target = torch.rand(R, 3, device=dev)
# vanilla
net = models.NeRF(num_coarse_samples=64, num_fine_samples=128).to(dev); net.load_state_dict(synth.vanilla_state(0))
opt = torch.optim.Adam(net.parameters(), lr=5e-4)
def vstep():
    opt.zero_grad(set_to_none=True)
    out = net(rays, True, False, 0.2, 3.0)
    (((out[0][0] - target) ** 2).mean() + ((out[1][0] - target) ** 2).mean()).backward(); opt.step()
timed("vanilla", vstep)
# PixelNeRF
batch = {k: v.to(dev) for k, v in cases.neo_batch(cases.strided_rays(R)).items()}
pix = models.PixelNeRF(num_src_views=cases.NV).to(dev); pix.load_state_dict(synth.pixelnerf_state(0))
latent = (torch.randn(cases.NV, 512, 240, 320, device=dev) * 0.1).requires_grad_(True)
pix.set_scene(latent, (640.0, 480.0))
popt = torch.optim.Adam(list(pix.parameters()) + [latent], lr=5e-4)
def pstep():
    popt.zero_grad(set_to_none=True)
    out = pix(batch, True, False, 0.2, 2.5)
    (((out[0][0] - target) ** 2).mean() + ((out[1][0] - target) ** 2).mean()).backward(); popt.step()
timed("pixelnerf", pstep)
pix.train_fused = False; timed("pix/layers", pstep); pix.train_fused = True     # the per-layer operators (rounds 4-5)
# Mip-NeRF 360
mrays = {k: v.to(dev) for k, v in cases.mip_rays(R).items()}
mip = models.MipNeRF360().to(dev); mip.load_state_dict(synth.mip360_state(0, weight_gain=0.5))
mopt = torch.optim.Adam(mip.parameters(), lr=5e-4)
def mstep():
    mopt.zero_grad(set_to_none=True)
    rend, hist = mip(mrays, 0.5, True, True, 0.2, 3.0)
    (((rend[2]["rgb"] - target) ** 2).mean() + 0.01 * sum((h["weights"] ** 2).sum(-1).mean() for h in hist)).backward(); mopt.step()
timed("mip360", mstep)
mip.train_fused = False; timed("mip/layers", mstep); mip.train_fused = True
