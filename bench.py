"""Headline benchmark: rays/s of the full-frame render path on synthetic 640x480
frames (BASELINE.json metric; workload = configs[1], vanilla NeRF 64 coarse + 128
fine samples per ray, random-init MLP), one process per GPU.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = ray generation for one frame + coarse and fine render of this rank's
contiguous ray range (+ one RCCL all-gather of the packed (rgb,depth,acc) tiles
when N > 1: the frame is fixed, so scaling is "strong").  Prints ONE JSON line on
rank 0.
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
N_COARSE, N_FINE = 64, 128
NEAR, FAR = 0.2, 3.0
FLOP_PER_POINT = 2 * 593408           # NeRFMLP MACs x 2 (vanilla_nerf/model.py:44-125)
POINTS_PER_RAY = (N_COARSE + 1) + (N_COARSE + 1 + N_FINE)
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md, dense fp32 MFMA


def cpu_baseline(state, rays_cpu, got_rgb, got_depth, budget_rays):
    """Oracle (CPU restatement of the reference, 'port') timed on a bounded sample of
    the same frame; also returns the parity of the GPU frame on those rays."""
    import oracle
    # 32 threads is the fastest setting of an 8..256 sweep on the GPU box's 2x64-core EPYC 9575F
    # (profiles/cpu_threads_r01.log: 625 rays/s at 32, 40 rays/s at 256 threads)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    sample = {k: v[:budget_rays] for k, v in rays_cpu.items()}
    t0 = time.perf_counter()
    rgb, depth = oracle.vanilla.render_chunked(state, sample, NEAR, FAR, chunk=1024)
    dt = time.perf_counter() - t0
    err_rgb = float((got_rgb[:budget_rays] - rgb).abs().max())
    err_depth = float((got_depth[:budget_rays] - depth).abs().max())
    mse = float(((got_rgb[:budget_rays].clamp(0, 1) - rgb.clamp(0, 1)) ** 2).mean())
    psnr = float("inf") if mse == 0 else -10.0 * torch.log10(torch.tensor(mse)).item()
    return dict(value=budget_rays / dt, unit="rays/s", cores=torch.get_num_threads(), kind="port",
                sample="first %d rays (%d reference chunks of 1024) of the same 640x480 frame, 64+128 samples, "
                       "torch fp32 on host cores, %.1f s" % (budget_rays, (budget_rays + 1023) // 1024, dt)), \
        dict(max_abs_rgb=err_rgb, max_abs_depth=err_depth, psnr_db=psnr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-rays", type=int, default=8192, help="rays in the CPU-baseline sample (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "launch one process per GPU (WORLD_SIZE=%d, --gpus %d)" % (world, args.gpus)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    torch.set_grad_enabled(False)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from neo360_amd import models, ops, synth
    from neo360_amd.parallel import shard_bounds, gather_tiles

    state = synth.vanilla_state(0)
    net = models.NeRF(num_coarse_samples=N_COARSE, num_fine_samples=N_FINE).to(dev)
    net.load_state_dict(state)
    c2w = synth.look_at_origin(40.0)
    R = H * W
    lo, hi = shard_bounds(R, world, rank, unit=1024)
    ctx = net._context(dev)

    def step():
        ro, vd, rd, _ = ops.get_ray_directions_and_rays(H, W, 0.8 * W, c2w, ctx=ctx)
        rays = dict(rays_o=ro[lo:hi], viewdirs=vd[lo:hi], rays_d=rd[lo:hi])
        res = net(rays, False, False, NEAR, FAR)
        tile = torch.cat([res[1][0], res[1][2][:, None], res[1][1][:, None]], dim=1)   # (r, 5)
        return gather_tiles(tile, R, world, unit=1024) if world > 1 else tile

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
    kern_ms, launches, points = ctx.read_timing()
    ctx.set_timing(False)
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        achieved = points * FLOP_PER_POINT / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
        out = {
            "metric": "rays/sec (128 samples/ray) + PSNR vs ref, 640x480",
            "value": R * args.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "vanilla_nerf 640x480 full frame, 64 coarse + 128 fine samples/ray "
                                   "(258 MLP points/ray), random-init 8x256 MLP, raygen + both levels"
                                   + (", rays sharded by 1024-ray chunks + RCCL all-gather of (rgb,depth,acc) tiles"
                                      if world > 1 else ""),
                       "rays_per_frame": R, "parallelism": "ray-shard x%d" % world},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                         "kernel": "k_vanilla_mlp", "launches": launches,
                         "avg_launch_ms": kern_ms / max(launches, 1),
                         "flop_per_point": FLOP_PER_POINT, "points_per_launch_avg": points / max(launches, 1)},
        }
        if world == 1 and args.cpu_rays > 0:
            ro, vd, rd, _ = ops.get_ray_directions_and_rays(H, W, 0.8 * W, c2w, ctx=ctx)
            n = min(args.cpu_rays, R)
            rays_cpu = dict(rays_o=ro[:n].cpu(), viewdirs=vd[:n].cpu(), rays_d=rd[:n].cpu())
            base, parity = cpu_baseline(state, rays_cpu, frame[:n, :3].cpu(), frame[:n, 3].cpu(), n)
            out["cpu_baseline"] = base
            out["parity_vs_cpu"] = parity
            out["speedup_vs_cpu"] = out["value"] / base["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
