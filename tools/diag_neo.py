"""GPU diagnostic: per-output max-abs error of the NeO-360 path vs fixtures (run on the GPU box)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases
from neo360_amd import models, synth
torch.set_grad_enabled(False)
DEV = "cuda"

def net_for(nc, nf, gain=1.0):
    net = models.NeRF_TP(num_coarse_samples=nc, num_fine_samples=nf, num_src_views=cases.NV).to(DEV)
    net.load_state_dict(synth.nerf_tp_state(0, density_gain=gain))
    sc = cases.small_scene()
    net.set_scene(sc["plane_xz"].to(DEV), sc["plane_xy"].to(DEV), sc["plane_yz"].to(DEV), sc["latent"].to(DEV), sc["image_wh"])
    return net

def run(tag, n, chunk, nc, nf, gain=1.0):
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(ROOT, "tests", "golden", "g4_neo_%s.npz" % tag)).items()}
    net = net_for(nc, nf, gain)
    b = {k: v.to(DEV) for k, v in cases.neo_batch(cases.strided_rays(n)).items()}
    outs = []
    for i in range(0, n, chunk):
        part = {k: (v if k.startswith("src_") else v[i:i + chunk]) for k, v in b.items()}
        outs.append(net(part, False, False, 0.0, 0.0, out_depth=True))
    cat = lambda lv, j: torch.cat([o[lv][j] for o in outs]).cpu()
    got = dict(rgb0=cat(0, 0), depth0=cat(0, 5), rgb1=cat(1, 0), fg1=cat(1, 1), bg1=cat(1, 2), fgacc1=cat(1, 3), lam1=cat(1, 4), depth1=cat(1, 5))
    line = []
    for k in got:
        e = (got[k].double() - g[k].double()).abs()
        line.append("%s max %.2e p99 %.2e" % (k, e.max(), e.flatten().quantile(0.99)))
    print(tag, "|", " | ".join(line), flush=True)

run("small", 300, 256, 32, 64)
run("sharp", 256, 256, 32, 64, 8.0)
run("1024", 1024, 1024, 128, 256)
run("1500", 1500, 1024, 128, 256)
