"""Error of the NeO-360 point evaluators against an fp64 evaluation of oracle.neo360.region_eval:
CPU fp32 oracle (= the reference's arithmetic), GPU k_tp_mlp (exact fp32 MFMA), GPU k_tp_mlp_h (split fp16, latent gathered
as the reference does), GPU k_tp_mlp_hp (split fp16, latent PRE-PROJECTED through the first-layer weights = the default).
Inside (slot 1) and outside (slot 3) the sphere; feature maps as they are, scaled x1e-4 and x1e3 (un-normalised encoder
outputs: the lo plane of the split loses bits below ~2^-25, the hi plane overflows at 65504).  Run on the GPU box;
output committed under profiles/."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases, oracle
from neo360_amd import _lib, models, synth
torch.set_grad_enabled(False)
torch.set_num_threads(16)
dev = "cuda"
R, N = 128, 200          # 25,600 points per slot, 51,200 per scale
params = synth.nerf_tp_state(0)
cb = cases.neo_batch(cases.strided_rays(R))
gb = {k: v.to(dev) for k, v in cb.items()}
far_c, _ = oracle.rays.sphere_exit_depth(cb["rays_o"], cb["rays_d"])
dbl = lambda d: {k: (v.double() if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in d.items()}
for scale in (1.0, 1e-4, 1e3):
    scene = {k: (v * scale if isinstance(v, torch.Tensor) else v) for k, v in cases.small_scene().items()}
    print("feature maps x %g  (std %.3g)" % (scale, float(scene["latent"].std())))
    for slot, prefix, inside in ((1, "fg_fine_mlp.", True), (3, "bg_fine_mlp.", False)):
        tv = (torch.linspace(0.03, 0.97, N)[None, :] * far_c) if inside else torch.linspace(0.99, 0.01, N)[None, :].expand(R, N).contiguous()
        rgb64, sig64 = oracle.neo360.region_eval(dbl(params), prefix, dbl(cb), dbl(scene), tv.double(), inside, far_c.double())
        rgb32, sig32 = oracle.neo360.region_eval(params, prefix, cb, scene, tv, inside, far_c)
        outs = {"cpu fp32 oracle": torch.cat([rgb32, sig32], -1)}
        for name, prec, pre in (("gpu f32 (k_tp_mlp)", "f32", True), ("gpu f16x3 noproj (k_tp_mlp_h)", "f16x3", False),
                                ("gpu f16x3 (k_tp_mlp_hp)", "f16x3", True)):
            net = models.NeRF_TP(num_coarse_samples=32, num_fine_samples=64, num_src_views=cases.NV).to(dev)
            net.precision = prec
            net.load_state_dict(params)
            net.set_scene(scene["plane_xz"].to(dev), scene["plane_xy"].to(dev), scene["plane_yz"].to(dev), scene["latent"].to(dev),
                          scene["image_wh"], preproject=pre)
            try:
                outs[name] = net.eval_mlp(slot, gb, tv.to(dev), far=far_c.to(dev)).cpu()
            except _lib.NeoError as e:
                outs[name] = None
            net.close()
        print("  slot %d (%s), %d points; error vs fp64 | relative sigma error uses max(|sigma|, 1)" % (slot, "inside" if inside else "outside", R * N))
        for name, o in outs.items():
            if o is None:
                print("    %-30s range guard raised (no result)" % name)
                continue
            er = (o[..., :3].double() - rgb64).abs()
            es = (o[..., 3:].double() - sig64).abs() / sig64.abs().clamp_min(1.0)
            print("    %-30s rgb max %.2e rms %.2e | sigma rel max %.2e rms %.2e" % (name, er.max(), er.pow(2).mean().sqrt(), es.max(), es.pow(2).mean().sqrt()))
