"""GPU: stage-by-stage comparison of the bg branch on the full-size C3 chunk against oracle intermediates
(tools/build/oracle_full_extra.npz, made by the CPU oracle in the build container), for chosen rays."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases
from neo360_amd import models, ops, synth
dev = "cuda"
RAYS = [int(x) for x in os.environ.get("RAYS", "962,817,819,23,100").split(",")]
net = models.NeRF_TP(num_coarse_samples=128, num_fine_samples=256, num_src_views=3).to(dev)
net.load_state_dict(synth.nerf_tp_state(0))
net.precision = os.environ.get("PREC", "f16x3")
sc = cases.full_scene()
net.set_scene(sc["plane_xz"].to(dev), sc["plane_xy"].to(dev), sc["plane_yz"].to(dev), sc["latent"].to(dev), sc["image_wh"])
gb = {k: v.to(dev) for k, v in cases.full_batch(1024).items()}
z = np.load(os.path.join(ROOT, "tools", "build", "oracle_full_extra.npz"))
T = lambda k: torch.from_numpy(z[k]).to(dev)
far = T("far_0")
def rep(name, got, want):
    e = (got - want).abs()
    while e.dim() > 1: e = e.amax(-1)
    print("%-28s max %.2e | " % (name, float(e.max())) + " ".join("r%d %.2e" % (r, float(e[r])) for r in RAYS))
# level 0 bg MLP at the oracle's positions
g0 = net.eval_mlp(2, gb, T("bg_s_0"), far=far, chunk=1024)
rep("bg coarse rgb", g0[..., :3], T("bg_rgb_0")); rep("bg coarse sigma", g0[..., 3:], T("bg_sigma_0"))
c0 = ops.composite(2, torch.cat([T("bg_rgb_0"), T("bg_sigma_0")], -1), T("bg_s_0"))
rep("bg coarse weights (oracle in)", c0["weights"], T("bg_w_0"))
s1 = ops.resample(T("bg_s_0"), T("bg_w_0"), 256, descending=True)
rep("bg resample (oracle in)", s1, T("bg_s_1"))
s1g = ops.resample(T("bg_s_0"), ops.composite(2, g0, T("bg_s_0"))["weights"], 256, descending=True)
rep("bg resample (gpu chain)", s1g, T("bg_s_1"))
g1 = net.eval_mlp(3, gb, T("bg_s_1"), far=far, chunk=1024)
rep("bg fine rgb", g1[..., :3], T("bg_rgb_1")); rep("bg fine sigma", g1[..., 3:], T("bg_sigma_1"))
c1 = ops.composite(2, torch.cat([T("bg_rgb_1"), T("bg_sigma_1")], -1), T("bg_s_1"))
rep("bg fine composite (oracle in)", c1["rgb"], T("bgc_1"))
c1g = ops.composite(2, g1, T("bg_s_1"))
rep("bg fine composite (gpu mlp)", c1g["rgb"], T("bgc_1"))
g1g = net.eval_mlp(3, gb, s1g, far=far, chunk=1024)
rep("bg fine composite (gpu chain)", ops.composite(2, g1g, s1g)["rgb"], T("bgc_1"))
for r in RAYS[:3]:
    d = (g1[r, :, 3] - T("bg_sigma_1")[r, :, 0]).abs()
    i = int(d.argmax())
    print("ray", r, "worst fine sigma at sample", i, "s=%.6f" % float(T("bg_s_1")[r, i]), "gpu %.6f ref %.6f" % (float(g1[r, i, 3]), float(T("bg_sigma_1")[r, i, 0])),
          "| rgb err there %.2e" % float((g1[r, i, :3] - T("bg_rgb_1")[r, i]).abs().max()))
    d = (g1[r, :, :3] - T("bg_rgb_1")[r]).abs().amax(-1)
    i = int(d.argmax())
    print("      worst fine rgb at sample", i, "s=%.6f err %.2e  weight there %.3e" % (float(T("bg_s_1")[r, i]), float(d[i]), float(T("bg_w_1")[r, i])))
