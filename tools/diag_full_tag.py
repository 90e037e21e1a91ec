"""GPU: per-ray error anatomy of a full-size C3 chunk (TAG = "", b1 .. b6) against its reference fixture, next to the reference's own
noise / margin / cdf self-displacement / flip size per ray; dumps the GPU outputs to gpurun_out/diag_<tag>/ for CPU-side analysis."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases
from neo360_amd import models, synth
dev = "cuda"
tag = os.environ.get("TAG", "b5")
nv = cases.FULL_B[tag]["nv"] if tag else 3
net = models.NeRF_TP(num_coarse_samples=128, num_fine_samples=256, num_src_views=nv).to(dev)
net.load_state_dict(synth.nerf_tp_state(0, density_gain=cases.full_gain(tag)))
sc, cb = cases.full_case(tag, 1024)
net.set_scene(sc["plane_xz"].to(dev), sc["plane_xy"].to(dev), sc["plane_yz"].to(dev), sc["latent"].to(dev), sc["image_wh"])
gb = {k: v.to(dev) for k, v in cb.items()}
name = "g4_neo_full" + ("_" + tag if tag else "")
g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
nz = np.load(os.path.join(ROOT, "tests", "golden", name + "_noise.npz"))
fl = np.load(os.path.join(ROOT, "tests", "golden", name + "_flip.npz"))
out = os.path.join(ROOT, "gpurun_out", "diag_" + (tag or "full")); os.makedirs(out, exist_ok=True)
for prec in ("f16x3", "f32"):
    net.precision = prec
    res = net(gb, False, False, 0.0, 0.0, out_depth=True)
    net.check_flags()
    got = dict(rgb0=res[0][0], depth0=res[0][5], rgb1=res[1][0], fg1=res[1][1], bg1=res[1][2], fgacc1=res[1][3], lam1=res[1][4], depth1=res[1][5])
    got = {k: v.cpu().numpy() for k, v in got.items()}
    np.savez_compressed(os.path.join(out, "gpu_%s.npz" % prec), **got)
    print("==== %s %s" % (name, prec))
    for k in ("rgb0", "rgb1", "bg1", "fg1", "depth1", "lam1", "fgacc1"):
        e = np.abs(got[k] - g[k]); e = e.max(-1) if e.ndim == 2 and e.shape[-1] == 3 else e.reshape(-1)
        top = np.argsort(-e)[:5]
        print(k, "max %.2e" % e.max(), "rays >= 5e-5:", int((e >= 5e-5).sum()))
        for i in top:
            print("    ray %4d err %.2e  noise %.2e  margin_bg %.2e  margin_fg %.2e  cdfnoise_bg %.2e  cdfnoise_fg %.2e  flip %.2e" % (
                i, e[i], nz["noise_" + k][i], nz["margin_bg1"][i], nz["margin_fg1"][i], nz["cdfnoise_bg1"][i] if "cdfnoise_bg1" in nz.files else -1,
                nz["cdfnoise_fg1"][i] if "cdfnoise_fg1" in nz.files else -1, fl["flip_" + k][i]))
