"""Which torch operators launch the strided copy / fill / add kernels of one NeO-360 training step (bench.py --workload neo360_train's
step): torch.profiler with shapes and Python stacks, top aten::copy_ / fill_ / add_ events by device time."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from neo360_amd import models, synth, training, ops
dev = "cuda"
torch.set_grad_enabled(True)
H, W = bench.H, bench.W
nv, B = 3, 500
net = models.NeRF_TP(num_coarse_samples=128, num_fine_samples=256, num_src_views=nv).to(dev)
net.load_state_dict(synth.nerf_tp_state(0))
sc = synth.scene_features(0, nv, 128, (120, 160), 512, (240, 320), std=0.1)
maps = [sc[k].to(dev).requires_grad_(True) for k in ("plane_xz", "plane_xy", "plane_yz", "latent")]
net.set_scene(*maps, (float(W), float(H)))
ro, vd, rd, _ = ops.get_ray_directions_and_rays(H, W, 0.8 * W, synth.look_at_origin(40.0))
sel = (torch.arange(B, device=dev) * 601 + 230 * W) % (H * W)
poses, focal, centre = synth.source_views(nv, W, H)
batch = dict(rays_o=ro[sel].contiguous(), rays_d=rd[sel].contiguous(), viewdirs=vd[sel].contiguous(), src_poses=poses.to(dev),
             src_focal=focal.to(dev), src_c=centre.to(dev), src_imgs=torch.zeros(nv, 3, H, W, device=dev))
target = synth.uniform(5, "train_target", (B, 3), 0.0, 1.0).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=5e-4)
interval = 1.0 / 385
def step(i):
    opt.zero_grad(set_to_none=True)
    for m in maps: m.grad = None
    lv = net(batch, True, False, 0.0, 0.0, out_depth=False, seed=1000 + i)
    loss = sum(((l[0] - target) ** 2).mean() for l in lv)
    loss = loss + 0.01 * (training.eff_distloss(lv[1][1], lv[1][3], interval) + training.eff_distloss(lv[1][2], lv[1][4], interval))
    loss.backward(); opt.step()
for i in range(3): step(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(10); torch.cuda.synchronize()
evs = [e for e in prof.events() if e.name in ("aten::copy_", "aten::fill_", "aten::add", "aten::add_", "aten::contiguous", "aten::clone", "aten::repeat", "aten::zeros", "aten::zero_")]
evs.sort(key=lambda e: -e.device_time_total)
for e in evs[:40]:
    st = [s for s in (e.stack or []) if "neo-360_amd" in s or "bench" in s or "prof_train" in s or "autograd" in s][:3]
    print("%-18s %8.1f us  %s  %s" % (e.name, e.device_time_total, e.input_shapes, " | ".join(s.split("/")[-1] for s in st)))
