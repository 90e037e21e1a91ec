"""Error of the three arithmetic paths of the vanilla MLP stage against an fp64 evaluation:
CPU fp32 oracle (= the reference's arithmetic), GPU exact-fp32 MFMA, GPU fp16-MFMA with hi/lo-split
operands.  Run on the GPU box; output committed under profiles/."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases, oracle
from neo360_amd import models, synth
torch.set_grad_enabled(False)
dev = "cuda"
for gain in (1.0, 8.0):
    state = synth.vanilla_state(0, density_gain=gain)
    rays = cases.strided_rays(256)
    t = torch.sort(synth.uniform(17, "ps_t", (256, 193), 0.2, 3.0), dim=-1).values
    def stage(dtype):
        p = {k: v.to(dtype) for k, v in state.items()}
        pts = oracle.sampling.points_on_rays(t, rays["rays_o"], rays["viewdirs"])       # fp32 points, as every path sees them
        enc = oracle.encoding.pos_enc(pts, 0, 10).to(dtype) if dtype == torch.float32 else oracle.encoding.pos_enc(pts.double(), 0, 10)
        de = oracle.encoding.pos_enc(rays["viewdirs"], 0, 4).to(dtype) if dtype == torch.float32 else oracle.encoding.pos_enc(rays["viewdirs"].double(), 0, 4)
        rgb, sig = oracle.mlp.vanilla_mlp(p, "fine_mlp.", enc, de)
        return torch.cat([oracle.mlp.colour_activation(rgb), oracle.mlp.density_activation(sig)], -1)
    truth = stage(torch.float64)
    cpu32 = stage(torch.float32)
    outs = {"cpu fp32 oracle": cpu32}
    for prec in ("f32", "f16x3"):
        net = models.NeRF().to(dev); net.load_state_dict(state); net.precision = prec
        outs["gpu " + prec] = net.eval_mlp(1, rays["rays_o"].to(dev), rays["viewdirs"].to(dev), t.to(dev)).cpu()
    print("density gain %.0f: per-point (rgb, sigma) error vs fp64, %d points" % (gain, truth.shape[0] * truth.shape[1]))
    for name, o in outs.items():
        e = (o.double() - truth).abs()
        print("  %-18s rgb max %.2e rms %.2e | sigma max %.2e rms %.2e" % (name, e[..., :3].max(), e[..., :3].pow(2).mean().sqrt(), e[..., 3].max(), e[..., 3].pow(2).mean().sqrt()))
