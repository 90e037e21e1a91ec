"""GPU diagnostic: run-to-run determinism and fp32-kernel agreement of every point evaluator, at scale.
  vanilla: 8192 rays x 128 points, split-fp16 vs fp32-MFMA kernel, 4 runs
  neo360 : R rays x 128 points fg/bg, split-fp16 vs fp32-MFMA kernel, 4 runs
"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases
from neo360_amd import models, synth, ops
torch.set_grad_enabled(False)
DEV = "cuda"
RUNS = 4


def report(name, ref, runs):
    errs = [(r - ref).abs() for r in runs]
    bad = [int((e.amax(-1) > 1e-4).sum()) for e in errs]
    same = all(torch.equal(runs[0], r) for r in runs[1:])
    print("%-12s points %8d  >1e-4 per run %s  bitwise-repeatable %s  max|split-f32| %.2e" %
          (name, ref[..., 0].numel(), bad, same, max(e.max().item() for e in errs)))


# ---- vanilla ----
R, N = int(os.environ.get("RV", "8192")), 128
g = torch.Generator().manual_seed(5)
o = (torch.rand(R, 3, generator=g) * 2 - 1).to(DEV)
d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(DEV)
t = (torch.rand(R, N, generator=g) * 4 + 2).sort(-1).values.to(DEV)
st = synth.vanilla_state(0)
nets = {}
for prec in ("f32", "f16x3"):
    net = models.NeRF().to(DEV)
    net.precision = prec
    net.load_state_dict(st)
    nets[prec] = net
for level in (0, 1):
    ref = nets["f32"].eval_mlp(level, o, d, t).cpu()
    again = nets["f32"].eval_mlp(level, o, d, t).cpu()
    runs = [nets["f16x3"].eval_mlp(level, o, d, t).cpu() for _ in range(RUNS)]
    report("vanilla L%d" % level, ref, runs)
    print("   fp32 kernel repeatable:", torch.equal(ref, again))

# ---- neo360 ----
R, NC = int(os.environ.get("RN", "2048")), 128
params = synth.nerf_tp_state(0)
scene = cases.small_scene()
batch = cases.neo_batch(cases.strided_rays(R))
gb = {k: v.to(DEV) for k, v in batch.items()}
far_g, _ = ops.intersect_sphere(gb["rays_o"], gb["rays_d"])
tv = torch.linspace(0.05, 0.95, NC, device=DEV)[None, :] * far_g.reshape(-1, 1)
tb = torch.linspace(0.02, 0.98, NC, device=DEV)[None, :].expand(R, NC).contiguous()


def mk(prec):
    net = models.NeRF_TP(num_coarse_samples=NC, num_fine_samples=256, num_src_views=cases.NV).to(DEV)
    net.precision = prec
    net.load_state_dict(params)
    net.set_scene(scene["plane_xz"].to(DEV), scene["plane_xy"].to(DEV), scene["plane_yz"].to(DEV), scene["latent"].to(DEV), scene["image_wh"])
    return net


ref_net, h_net = mk("f32"), mk("f16x3")
for name, slot, tt in (("neo fg0", 0, tv), ("neo fg1", 1, tv), ("neo bg0", 2, tb), ("neo bg1", 3, tb)):
    ref = ref_net.eval_mlp(slot, gb, tt, far=far_g).cpu()
    again = ref_net.eval_mlp(slot, gb, tt, far=far_g).cpu()
    runs = [h_net.eval_mlp(slot, gb, tt, far=far_g).cpu() for _ in range(RUNS)]
    report(name, ref, runs)
    print("   fp32 kernel repeatable:", torch.equal(ref, again))

# ---- pixelnerf ----
R, NC = int(os.environ.get("RP", "2048")), 129
pnet = models.PixelNeRF(num_src_views=cases.NV).to(DEV)
pnet.load_state_dict(synth.pixelnerf_state(0))
pnet.set_scene(scene["latent"].to(DEV), scene["image_wh"])
batch = cases.neo_batch(cases.strided_rays(R))
gb = {k: v.to(DEV) for k, v in batch.items()}
tt = torch.linspace(0.2, 2.5, NC, device=DEV)[None, :].expand(R, NC).contiguous()
for slot in (0, 1):
    runs = [pnet.eval_mlp(slot, gb, tt).cpu() for _ in range(RUNS)]
    same = all(torch.equal(runs[0], r) for r in runs[1:])
    d = max((runs[0] - r).abs().max().item() for r in runs[1:])
    print("pixelnerf slot %d points %8d  bitwise-repeatable %s  max run-to-run diff %.2e" % (slot, R * NC, same, d))
