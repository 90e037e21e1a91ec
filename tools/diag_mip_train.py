"""Where do the gradients of the Mip-NeRF 360 training chain differ from fp64 autograd: the encodings' fp32 conditioning (ReLU units
within reach of a kink flip their derivative) or the chain's arithmetic?  Compares the library's gradients with fp64 / fp32 autograd
of the oracle (a) on the oracle's own encodings, (b) on the library's encodings."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import cases
from oracle import mip360
from neo360_amd import models, synth, training
DEV = "cuda"
R, counts = 96, (16, 8)
net = models.MipNeRF360(num_prop_samples=counts[0], num_nerf_samples=counts[1]).to(DEV)
sd = synth.mip360_state(0, weight_gain=0.25)
net.load_state_dict(sd)
rays_c = cases.mip_rays(R); rays = {k: v.to(DEV) for k, v in rays_c.items()}
target = synth.uniform(93, "mip_target", (R, 3), 0.0, 1.0)
probes = [synth.uniform(94 + l, "mip_probe", (R, n), 0.0, 1.0) for l, n in enumerate((counts[0], counts[0], counts[1]))]
names = sorted(k for k in sd if not k.endswith("pos_basis_t"))
for p in net.parameters(): p.requires_grad_(True)
rend, hist = training.mip_render_train(net, rays, 0.5, True, 0.2, 3.0, seed=13)
loss_g = ((rend[2]["rgb"] - target.to(DEV)) ** 2).mean() + 0.05 * sum((h["weights"] * p.to(DEV)).sum(-1).mean() for h, p in zip(hist, probes))
params = dict(net.named_parameters())
g_g = [g.detach().cpu().double() for g in torch.autograd.grad(loss_g, [params[k] for k in names])]
sdists = [h["sdist"].detach().cpu() for h in hist]
near, far = 0.2, 3.0
x0s = []
for sdv, n in zip(sdists, (counts[0], counts[0], counts[1])):
    td = 1 / (sdv * (1 / far) + (1 - sdv) * (1 / near))
    x0s.append(training.mip_encode(rays["rays_o"], rays["rays_d"], rays["radii"], td.to(DEV), net.mlps[0].pos_basis_t).cpu().reshape(R, n, 504))
def oracle_grads(dtype, x0):
    cv = lambda v: v.to(dtype) if torch.is_floating_point(v) else v
    pp = {k: (cv(v).clone().requires_grad_(True) if k in names else cv(v)) for k, v in sd.items()}
    r2, h2 = mip360.render(pp, {k: cv(v) for k, v in rays_c.items()}, 0.5, 0.2, 3.0, num_prop_samples=counts[0], num_nerf_samples=counts[1],
                           sdist_given=[cv(x) for x in sdists], basis=cv(mip360.icosahedron_basis()), x0_given=None if x0 is None else [cv(x) for x in x0])
    loss = ((r2[2]["rgb"] - cv(target)) ** 2).mean() + 0.05 * sum((h["weights"] * cv(p)).sum(-1).mean() for h, p in zip(h2, probes))
    return [g.double() for g in torch.autograd.grad(loss, [pp[k] for k in names])]
a64, a32 = oracle_grads(torch.float64, None), oracle_grads(torch.float32, None)
b64, b32 = oracle_grads(torch.float64, x0s), oracle_grads(torch.float32, x0s)
rel = lambda x, ref: float((x - ref).norm()) / (float(ref.norm()) + 1e-30)
print("%-34s %10s %10s | %10s %10s | %10s" % ("tensor", "lib-a64", "a32-a64", "lib-b64", "b32-b64", "a64-b64"))
for nm, g, p, q, r, s in zip(names, g_g, a64, a32, b64, b32):
    print("%-34s %10.2e %10.2e | %10.2e %10.2e | %10.2e" % (nm, rel(g, p), rel(q, p), rel(g, r), rel(s, r), rel(p, r)))
