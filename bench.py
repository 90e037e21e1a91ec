"""Headline benchmark: rays/s of the full-frame render path on synthetic 640x480
frames (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus 1] [--steps K] [--warmup W] [--workload vanilla|neo360]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Default workload = BASELINE.json configs[1] (the configuration the metric is quoted on):
vanilla NeRF, 640x480, 64 coarse + 128 fine samples/ray, random-init MLP.
`--workload neo360` = configs[2]/[3]: the NeO-360 tri-planar decoder, 3 source views,
128 coarse + 256 fine samples, inside + outside sphere, reference chunk 1024.

A step = ray generation for one frame + coarse and fine render of this rank's contiguous
range of whole 1024-ray chunks (+ ONE RCCL all-gather of the packed (rgb,depth,acc) tiles
when N > 1).  The frame is fixed, so scaling is "strong".  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W = 480, 640
CHUNK = 1024
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: dense fp32 MFMA (= fp32 vector peak)
PEAK_F16_MFMA_TFLOPS = 2500.0         # MI355X_MICROARCH.md: dense fp16/bf16 MFMA
CPU_THREADS = 32                      # fastest of an 8..256 sweep on the GPU box (profiles/cpu_threads_r01.log)


def build_vanilla(dev):
    from neo360_amd import models, synth
    state = synth.vanilla_state(0)
    net = models.NeRF(num_coarse_samples=64, num_fine_samples=128).to(dev)
    net.load_state_dict(state)
    extra = {}
    desc = ("vanilla_nerf 640x480 full frame, 64 coarse + 128 fine samples/ray (65+193 = 258 MLP points/ray), "
            "random-init 8x256 MLP, raygen + both levels")
    return net, state, extra, None, desc, dict(near=0.2, far=3.0), "k_vanilla_mlp", 8192


def build_neo360(dev):
    from neo360_amd import models, synth
    nv = 3
    state = synth.nerf_tp_state(0)
    net = models.NeRF_TP(num_coarse_samples=128, num_fine_samples=256, num_src_views=nv).to(dev)
    net.load_state_dict(state)
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    # stand-ins for the scene encoder's outputs, reference shapes: N(0, 0.1^2)
    scene = {k: torch.randn(nv, 128, 120, 160, device=dev, generator=g) * 0.1 for k in ("plane_xz", "plane_xy", "plane_yz")}
    scene["latent"] = torch.randn(nv, 512, 240, 320, device=dev, generator=g) * 0.1
    scene["image_wh"] = (float(W), float(H))
    net.set_scene(scene["plane_xz"], scene["plane_xy"], scene["plane_yz"], scene["latent"], scene["image_wh"])
    poses, focal, centre = synth.source_views(nv, W, H)
    extra = dict(src_poses=poses.to(dev), src_focal=focal.to(dev), src_c=centre.to(dev),
                 src_imgs=torch.zeros(nv, 3, H, W, device=dev))
    desc = ("neo360 tri-planar decoder 640x480 full frame, 3 source views, 128 coarse + 256 fine samples/ray, "
            "inside + outside sphere ((129+385)x2 = 1028 MLP points/ray x 3 views), reference chunk 1024, "
            "random-init MLPs, synthetic N(0,0.1) tri-planes (3x128x120x160) + latents (3x512x240x320)")
    return net, state, extra, scene, desc, dict(near=0.0, far=0.0), "k_tp_mlp", 256


def build_pixelnerf(dev):
    from neo360_amd import models, synth
    nv = 3
    state = synth.pixelnerf_state(0)
    net = models.PixelNeRF(num_coarse_samples=64, num_fine_samples=64, num_src_views=nv).to(dev)
    net.load_state_dict(state)
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    scene = {"latent": torch.randn(nv, 512, 240, 320, device=dev, generator=g) * 0.1, "image_wh": (float(W), float(H))}
    net.set_scene(scene["latent"], scene["image_wh"])
    poses, focal, centre = synth.source_views(nv, W, H)
    extra = dict(src_poses=poses.to(dev), src_focal=focal.to(dev), src_c=centre.to(dev),
                 src_imgs=torch.zeros(nv, 3, H, W, device=dev))
    desc = ("PixelNeRF baseline decoder 640x480 full frame, 3 source views, 64 coarse + 64 fine samples/ray "
            "((65+129) MLP points/ray x 3 views), reference chunk 1024, random-init MLPs, synthetic N(0,0.1) "
            "latents (3x512x240x320)")
    return net, state, extra, scene, desc, dict(near=0.2, far=3.0), "k_pix_mlp", 1024


def build_mip360(dev, n_nerf=32):
    from neo360_amd import models, synth
    state = synth.mip360_state(0, weight_gain=0.5)
    net = models.MipNeRF360(num_prop_samples=64, num_nerf_samples=n_nerf).to(dev)
    net.load_state_dict(state)
    desc = ("mipnerf360 640x480 full frame, 2 proposal levels x 64 samples (PropMLP 4x256) + %d NeRF samples "
            "(NeRFMLP 8x1024), cone casting + contraction + 504-d IPE, random-init MLPs (kaiming x0.5)" % n_nerf)
    return net, state, {}, None, desc, dict(near=0.2, far=3.0, train_frac=1.0), "k_mip_mlp", (4096 if n_nerf == 128 else 8192)


def cpu_baseline(workload, state, scene, rays_cpu, extra, kw, n):
    """The oracle (CPU restatement of the reference: kind 'port') timed on a bounded sample
    of the same frame on the host cores."""
    import oracle
    torch.set_num_threads(min(CPU_THREADS, os.cpu_count() or 1))
    sample = {k: v[:n] for k, v in rays_cpu.items()}
    t0 = time.perf_counter()
    if workload == "vanilla":
        rgb, depth = oracle.vanilla.render_chunked(state, sample, kw["near"], kw["far"], chunk=CHUNK)
    elif workload == "pixelnerf":
        batch = dict(sample)
        batch.update({k: v.cpu() for k, v in extra.items()})
        sc = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in scene.items()}
        rgb, depth = oracle.pixelnerf.render_chunked(state, batch, sc, kw["near"], kw["far"], chunk=CHUNK)
    elif workload.startswith("mip360"):
        from oracle import mip360
        rend, _ = mip360.render(state, sample, kw["train_frac"], kw["near"], kw["far"], num_prop_samples=64,
                                num_nerf_samples=128 if workload.endswith("128") else 32)
        rgb, depth = rend[-1]["rgb"], torch.zeros(n)
    else:
        batch = dict(sample)
        batch.update({k: v.cpu() for k, v in extra.items()})
        sc = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in scene.items()}
        rgb, depth = oracle.neo360.render_chunked(state, batch, sc, chunk=n)
    dt = time.perf_counter() - t0
    base = dict(value=n / dt, unit="rays/s", cores=torch.get_num_threads(), kind="port",
                sample="first %d rays of the same 640x480 frame as one%s reference chunk%s, same weights / features, "
                       "torch fp32 CPU oracle, %.1f s" % (n, "" if n <= CHUNK else " run of", "" if n <= CHUNK else "s", dt))
    return base, rgb, depth


def pmc_traffic(workload, precision):
    """HBM bytes per dominant-kernel launch from the committed rocprofv3 PMC passes of this same command
    (tools/pmc_bench.sh -> profiles/r01_pmc_<workload>_<precision>.json): (2 x FETCH_SIZE + WRITE_SIZE) KiB,
    mean over the launches of a frame.  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950
    (wide coalesced reads are tallied at half their bytes).  None when no profile is committed."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_%s_%s.json" % (workload, precision))
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f)["derived"].get("hbm_bytes_per_launch")
    if workload == "vanilla" and precision == "f32":
        path = os.path.join(ROOT, "profiles", "r01_vanilla_pmc_summary.json")
        if os.path.exists(path):
            with open(path) as f:
                disp = json.load(f)["dispatches"]
            per = [(2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0 for d in disp.values() if "FETCH_SIZE" in d and "WRITE_SIZE" in d]
            return sum(per) / len(per) if per else None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=("vanilla", "neo360", "pixelnerf", "mip360", "mip360_128"), default="vanilla")
    ap.add_argument("--precision", choices=("auto", "f32", "f16x3"), default="auto",
                    help="MLP arithmetic: exact fp32 MFMA, or fp16 MFMA with hi/lo-split operands (fp32-equivalent, "
                         "the default of every renderer)")
    ap.add_argument("--cpu-rays", type=int, default=-1, help="rays in the CPU-baseline sample (0 = skip, -1 = default)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "launch one process per GPU (WORLD_SIZE=%d, --gpus %d)" % (world, args.gpus)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    torch.set_grad_enabled(False)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from neo360_amd import ops, render, synth

    if args.workload == "vanilla":
        built = build_vanilla(dev)
    elif args.workload == "neo360":
        built = build_neo360(dev)
    elif args.workload == "pixelnerf":
        built = build_pixelnerf(dev)
    else:       # reference defaults (64,64,32), or BASELINE.json's wording "64 proposal + 128 fine"
        built = build_mip360(dev, 128 if args.workload.endswith("128") else 32)
    net, state, extra, scene, desc, kw, kernel_name, cpu_default = built
    if args.precision == "auto":
        args.precision = getattr(net, "default_precision", "f32")
    split = args.precision == "f16x3"
    net.precision = args.precision
    kernel_name = kernel_name + "_h" if split else kernel_name
    c2w = synth.look_at_origin(40.0)
    R = H * W
    ctx = net._context(dev)

    def frame_rays():
        ro, vd, rd, radii = ops.get_ray_directions_and_rays(H, W, 0.8 * W, c2w, ctx=ctx)
        batch = dict(rays_o=ro, viewdirs=vd, rays_d=rd)
        if args.workload.startswith("mip360"):
            batch["radii"] = radii[:, None]
        batch.update(extra)
        return batch

    def step():
        return render.render_frame_sharded(net, frame_rays(), world, rank, chunk=CHUNK, **kw)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame = step()
    fence()
    dt = time.perf_counter() - t0
    kern_ms, launches, points, flops = ctx.read_timing()
    ctx.set_timing(False)
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        achieved = flops / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
        alg_bytes_per_point = 20.0 + {"neo360": 3 * 14336.0, "pixelnerf": 3 * 8192.0}.get(args.workload, 0.0)
        # split path: every algorithmic product costs three fp16 MFMA products, so the ceiling for
        # ALGORITHMIC flops on the fp16 pipe is peak/3
        peak = PEAK_F16_MFMA_TFLOPS / 3.0 if split else PEAK_F32_MFMA_TFLOPS
        out = {
            "metric": "rays/sec (128 samples/ray) + PSNR vs ref, 640x480",
            "value": R * args.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 (fp16-MFMA products of hi/lo-split fp32 operands, fp32 accumulate)" if split else "f32",
            "data": "synthetic",
            "config": {"workload": desc + (", rays sharded by whole 1024-ray chunks + one RCCL all-gather of "
                                           "(rgb,depth,acc) tiles" if world > 1 else ""),
                       "rays_per_frame": R, "parallelism": "ray-shard x%d" % world},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": pmc_traffic(args.workload, args.precision), "kernel": kernel_name,
                         "launches": launches, "avg_launch_ms": kern_ms / max(launches, 1),
                         "algorithmic_flop_per_launch": flops / max(launches, 1),
                         "points_per_launch": points / max(launches, 1),
                         "algorithmic_bytes_per_launch": points / max(launches, 1) * alg_bytes_per_point,
                         "peak_definition": ("dense fp16 MFMA peak 2500 TFLOP/s / 3 products per algorithmic multiply "
                                             "(a_hi*b_hi + a_hi*b_lo + a_lo*b_hi); executed matrix rate = 3 x achieved; "
                                             "the exact-fp32-MFMA kernel (--precision f32) peaks at 157.3")
                         if split else "dense fp32 MFMA peak",
                         "note": "rank 0's launches; algorithmic flops = reference formulation MACs x 2 (SURVEY.md 8d); "
                                 "traffic = HBM bytes/launch from the committed PMC passes (profiles/); algorithmic "
                                 "bytes = 4 B t in + 16 B (rgb,sigma) out per point" +
                                 (" + 3 views x 14,336 B of feature taps per point as the reference gathers them (no "
                                  "reuse; SURVEY.md 8d upper bound; every texel once would be 560 MB per frame)"
                                  if args.workload == "neo360" else "")},
        }
        n_cpu = cpu_default if args.cpu_rays < 0 else args.cpu_rays
        if world == 1 and n_cpu > 0:
            batch = frame_rays()
            n = min(n_cpu, R)
            rays_cpu = {k: batch[k][:n].cpu() for k in ("rays_o", "viewdirs", "rays_d", "radii") if k in batch}
            base, rgb_c, depth_c = cpu_baseline(args.workload, state, scene, rays_cpu, extra, kw, n)
            if args.workload in ("neo360",) and n != CHUNK:
                # NeO-360 results depend on chunk membership: render the same rays as their own chunk
                sub = {k: (v[:n] if k in ("rays_o", "viewdirs", "rays_d") else v) for k, v in batch.items()}
                got = render.render_rays_test(net, sub, chunk=n, **kw)
                rgb_g, depth_g = got["rgb"].cpu(), got["depth"].cpu()
            else:
                rgb_g, depth_g = frame[:n, :3].cpu(), frame[:n, 3].cpu()
            err = (rgb_g - rgb_c).abs().amax(dim=-1)
            out["cpu_baseline"] = base
            out["parity_vs_cpu"] = {"max_abs_rgb": float(err.max()), "p99_abs_rgb": float(err.quantile(0.99)),
                                    "max_abs_depth": float((depth_g - depth_c).abs().max()),
                                    "psnr_db": render.psnr(rgb_g, rgb_c), "rays": n}
            out["speedup_vs_cpu"] = out["value"] / base["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
