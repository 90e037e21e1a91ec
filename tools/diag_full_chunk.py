"""GPU: per-ray error anatomy of the full-size C3 chunk (fixture g4_neo_full) - which rays, which outputs, next to the
reference's own noise estimates (fixture g4_neo_full_noise).  Dumps the GPU outputs to gpurun_out/ for CPU-side analysis."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases
from neo360_amd import models, synth
dev = "cuda"
net = models.NeRF_TP(num_coarse_samples=128, num_fine_samples=256, num_src_views=3).to(dev)
net.load_state_dict(synth.nerf_tp_state(0))
net.precision = os.environ.get("PREC", "f16x3")
sc = cases.full_scene()
net.set_scene(sc["plane_xz"].to(dev), sc["plane_xy"].to(dev), sc["plane_yz"].to(dev), sc["latent"].to(dev), sc["image_wh"])
gb = {k: v.to(dev) for k, v in cases.full_batch(1024).items()}
res = net(gb, False, False, 0.0, 0.0, out_depth=True)
net.check_flags()
got = dict(rgb0=res[0][0], depth0=res[0][5], rgb1=res[1][0], fg1=res[1][1], bg1=res[1][2], fgacc1=res[1][3], lam1=res[1][4], depth1=res[1][5])
got = {k: v.cpu().numpy() for k, v in got.items()}
g = np.load(os.path.join(ROOT, "tests", "golden", "g4_neo_full.npz"))
nz = np.load(os.path.join(ROOT, "tests", "golden", "g4_neo_full_noise.npz"))
out = os.path.join(ROOT, "gpurun_out", "r03e"); os.makedirs(out, exist_ok=True)
np.savez_compressed(os.path.join(out, "gpu_full_chunk_%s.npz" % net.precision), **got)
for k in ("rgb1", "bg1", "fg1", "depth1", "lam1"):
    e = np.abs(got[k] - g[k]); e = e.max(-1) if e.ndim == 2 and e.shape[-1] == 3 else e.reshape(-1)
    top = np.argsort(-e)[:6]
    print(k, "top rays:", [(int(i), "%.2e" % e[i], "noise %.2e" % nz["noise_" + k][i], "n64 %.2e" % (nz["noise64_" + k][i] if "noise64_" + k in nz.files else -1)) for i in top])
