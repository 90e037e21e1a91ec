"""Outputs of the exact-fp32 NeO-360 evaluator on a small seeded scene, every MLP slot, whole-batch and chunked direction tiling
(quirk Q1: chunk < rays), saved to $OUT; run once per library ($NEO360_HIP_LIB) and compare with `--compare a.pt b.pt` (bitwise).
Used for the A/B of the per-ray direction-sum table in k_tp_mlp (tools/gpu_r06ds.sh)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) == 4 and sys.argv[1] == "--compare":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    ok = set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)
    worst = max(float((a[k] - b[k]).abs().max()) for k in a)
    print("bitwise equal: %s (%d tensors, max |diff| %.3g, nonzero outputs: %s)" % (ok, len(a), worst, all(bool(a[k].abs().sum() > 0) for k in a)))
    sys.exit(0 if ok else 1)
from neo360_amd import models, synth, ops
torch.set_grad_enabled(False)
dev = torch.device("cuda")
NV, R = 3, 256
net = models.NeRF_TP(num_src_views=NV).to(dev)
net.precision = "f32"
net.load_state_dict(synth.nerf_tp_state(0))
H, W, focal = 480, 640, 512.0
g = torch.Generator(device=dev); g.manual_seed(0)
planes = [torch.randn(NV, 128, 120, 160, device=dev, generator=g) * 0.1 for _ in range(3)]
latent = torch.randn(NV, 512, 240, 320, device=dev, generator=g) * 0.1
net.set_scene(planes[0], planes[1], planes[2], latent, (float(W), float(H)))
ro, vd, rd, _ = ops.get_ray_directions_and_rays(H, W, focal, synth.look_at_origin(40.0))
sel = torch.randperm(H * W, device=dev, generator=torch.Generator(device=dev).manual_seed(1))[:R]
poses, sfocal, centre = synth.source_views(NV, W, H)
rays = {"rays_o": ro[sel].contiguous(), "rays_d": rd[sel].contiguous(), "viewdirs": vd[sel].contiguous(),
        "src_poses": poses.to(dev), "src_focal": sfocal.to(dev), "src_c": centre.to(dev)}
far, _ = ops.intersect_sphere(rays["rays_o"], rays["rays_d"])
out = {}
for slot in range(4):
    N = 33 if slot in (1, 3) else 17
    if slot < 2:
        t = torch.linspace(0.02, 0.98, N, device=dev)[None, :] * far.reshape(-1, 1)
    else:
        t = torch.linspace(0.98, 0.02, N, device=dev)[None, :].expand(R, N).contiguous()
    for chunk in (R, 64, 48):
        out["slot%d_chunk%d" % (slot, chunk)] = net.eval_mlp(slot, rays, t, far=far, chunk=chunk).cpu()
torch.save(out, os.environ["OUT"])
print("saved %d tensors; whole-batch vs chunk 64 colours differ: %s" % (len(out), not torch.equal(out["slot0_chunk%d" % R], out["slot0_chunk64"])))
