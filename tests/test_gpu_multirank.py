"""GPU: the multi-GPU path on the hardware a gpurun box has - ONE MI355X (VERDICT r3 task 3).

(a) RCCL itself: a one-rank "nccl" process group on the device; `parallel.gather_tiles` runs
    `all_gather_into_tensor` on device tiles (contiguous, non-contiguous -> padded send buffer, `out=`, `reuse=`).
    The reference's collective: models/interface.py:30-50 (`alter_gather_cat`), called from neo360/model.py:1073-1086.
(b) the sharding logic with the REAL renderer: two ranks share the one GPU (gloo process group, host-staged tiles:
    RCCL refuses two ranks on one device), each renders its `shard_bounds` range of a NeO-360 frame whose last chunk
    is short, and the assembled frame is bitwise the single-process frame.
(c) `bench.py --gpus 1` under torch.distributed.run with one rank: init_process_group("nccl"), barrier, max-reduce
    and the tile all-gather of the N-GPU job all execute; the line keeps the driver's contract.
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env(port, rank, world):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")


def _rccl_world1(rank, port, ret):
    import torch.distributed as dist
    from neo360_amd import parallel
    _env(port, 0, 1)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        ok = dist.get_backend() == "nccl"
        n = 1500
        frame = torch.arange(n * 5, dtype=torch.float32, device=dev).reshape(n, 5)
        got = parallel.gather_tiles(frame.clone(), n, 1, unit=1024)
        ok = ok and got.is_cuda and bool(torch.equal(got, frame)) and got.data_ptr() != frame.data_ptr()
        # a non-contiguous tile goes through the padded send buffer
        wide = torch.arange(n * 10, dtype=torch.float32, device=dev).reshape(n, 10)
        view = wide[:, ::2]
        got = parallel.gather_tiles(view, n, 1, unit=1024)
        ok = ok and bool(torch.equal(got, view.contiguous()))
        mine = torch.empty(n, 5, device=dev)
        ok = ok and parallel.gather_tiles(frame.clone(), n, 1, unit=1024, out=mine) is mine and bool(torch.equal(mine, frame))
        a = parallel.gather_tiles(frame.clone(), n, 1, unit=1024, reuse=True)
        b = parallel.gather_tiles(frame + 2.0, n, 1, unit=1024, reuse=True)
        ok = ok and a.data_ptr() == b.data_ptr() and bool(torch.equal(b, frame + 2.0))
        # barrier + max-reduce as bench.py issues them
        dist.barrier()
        t = torch.tensor([3.5], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and float(t.item()) == 3.5
        torch.cuda.synchronize()
        ret["ok"] = ok
    finally:
        dist.destroy_process_group()


def test_rccl_all_gather_executes_on_device():
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_rccl_world1, args=(port, ret), nprocs=1, join=True)
        assert ret.get("ok") is True, dict(ret)


N_RAYS, CHUNK = 2 * 1024 + 300, 1024          # three reference chunks, the last one short: shards of 2048 and 300 rays


def _model(dev):
    from neo360_amd import models, synth
    net = models.NeRF_TP(num_coarse_samples=16, num_fine_samples=32, num_src_views=cases.NV).to(dev)
    net.load_state_dict(synth.nerf_tp_state(0))
    sc = cases.small_scene()
    net.set_scene(sc["plane_xz"].to(dev), sc["plane_xy"].to(dev), sc["plane_yz"].to(dev), sc["latent"].to(dev), sc["image_wh"])
    return net


def _frame_batch(dev):
    return {k: v.to(dev) for k, v in cases.neo_batch(cases.strided_rays(N_RAYS)).items()}


def _two_ranks_one_gpu(rank, world, port, ret):
    import torch.distributed as dist
    from neo360_amd import parallel, render
    _env(port, rank, world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_grad_enabled(False)
        net, batch = _model(dev), _frame_batch(dev)
        lo, hi = parallel.shard_bounds(N_RAYS, world, rank, unit=CHUNK)
        tile = render.render_frame_sharded(net, batch, world, rank, chunk=CHUNK, gather=False)      # the real HIP renderer
        assert tile.is_cuda and tile.shape == (hi - lo, 5)
        frame = parallel.gather_tiles(tile.cpu(), N_RAYS, world, unit=CHUNK)                        # host-staged exchange
        ret[rank] = frame
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_assemble_the_single_process_frame():
    from neo360_amd import parallel, render
    dev = torch.device("cuda", 0)
    whole = render.render_frame_sharded(_model(dev), _frame_batch(dev), 1, 0, chunk=CHUNK)           # world 1: the whole frame
    assert whole.shape == (N_RAYS, 5)
    assert parallel.shard_counts(N_RAYS, 2, CHUNK) == [2048, 300]
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_two_ranks_one_gpu, args=(2, port, ret), nprocs=2, join=True)
        for r in range(2):
            assert torch.equal(ret[r], whole.cpu()), "rank %d: assembled frame differs from the single-process frame" % r


def test_bench_under_torchrun_one_rank():
    """The driver's N > 1 command line with N = 1: RCCL initialises, the collectives run, the contract holds."""
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
           "--cpu-rays", "0", "--others", "0", "--exact-f32", "0", "--setup-timing", "0"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 1 and out["unit"] == "rays/s" and out["value"] > 1e4
    assert "RCCL process group of 1 rank" in out["config"]["collective"]
    assert out["roofline"]["all_evaluator_launches"]["launches"] == 4      # inside / outside the sphere x coarse / fine of the one timed frame
    assert sum(k["launches"] for k in out["roofline"]["kernels"].values()) == 4
    path = os.path.join(ROOT, "gpurun_out", "bench_torchrun_world1.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(lines[0] + "\n")
