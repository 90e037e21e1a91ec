"""CPU: the ray-sharding / tile all-gather layer.  world_size-2 (and 3) runs over the
gloo backend stand in for RCCL: same torch.distributed calls, same code path."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from neo360_amd import models, parallel


@pytest.mark.parametrize("n,world,unit", [(307200, 8, 1024), (307200, 1, 1024), (1500, 2, 1024), (1500, 4, 1024),
                                          (300, 2, 256), (5, 3, 2), (0, 2, 1024), (1024, 8, 1024)])
def test_shard_counts_cover_whole_chunks(n, world, unit):
    counts = parallel.shard_counts(n, world, unit)
    assert len(counts) == world and sum(counts) == n
    # every shard except possibly the one holding the short last chunk is a whole number of chunks,
    # and shards are contiguous: concatenating them in rank order reproduces the frame
    lo = 0
    for r in range(world):
        a, b = parallel.shard_bounds(n, world, r, unit)
        assert a == lo and b - a == counts[r]
        assert a % unit == 0 or counts[r] == 0      # shards start on a chunk boundary (empty trailing shards aside)
        lo = b
    assert lo == n
    if n == 307200 and world == 8:
        assert counts == [38 * 1024] * 4 + [37 * 1024] * 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, unit, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frame = torch.arange(n * 5, dtype=torch.float32).reshape(n, 5)     # stand-in for (rgb, depth, acc)
        lo, hi = parallel.shard_bounds(n, world, rank, unit)
        got = parallel.gather_tiles(frame[lo:hi].clone(), n, world, unit)
        ok = bool(torch.equal(got, frame))
        # a second frame must not overwrite the first: fresh tensor by default ...
        got2 = parallel.gather_tiles((frame[lo:hi] + 1.0).clone(), n, world, unit)
        ok = ok and bool(torch.equal(got, frame)) and bool(torch.equal(got2, frame + 1.0)) and got2.data_ptr() != got.data_ptr()
        # ... `out=` fills the caller's tensor, `reuse=True` hands out the cached one
        mine = torch.empty(n, 5)
        got3 = parallel.gather_tiles(frame[lo:hi].clone(), n, world, unit, out=mine)
        ok = ok and got3 is mine and bool(torch.equal(mine, frame))
        a = parallel.gather_tiles(frame[lo:hi].clone(), n, world, unit, reuse=True)
        b = parallel.gather_tiles((frame[lo:hi] + 2.0).clone(), n, world, unit, reuse=True)
        ok = ok and a.data_ptr() == b.data_ptr() and bool(torch.equal(b, frame + 2.0))
        parallel.clear_buffers()
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,world,unit", [(1500, 2, 1024), (2048, 2, 1024), (700, 3, 256)])
def test_gather_tiles_gloo(n, world, unit):
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, n, unit, ret), nprocs=world, join=True)
        assert all(ret.get(r) for r in range(world)), dict(ret)


def test_psnr_formula():
    from neo360_amd import render
    a = torch.full((4, 3), 0.5)
    assert render.psnr(a, a) == float("inf")
    b = a + 0.1
    assert abs(render.psnr(b, a) - 20.0) < 1e-4     # mse = 0.01 -> 20 dB
    # inputs are clipped to [0,1] before the mse, as the reference does
    assert render.psnr(torch.full((2, 3), 2.0), torch.full((2, 3), 1.0)) == float("inf")


class _FakePixelNeRF(models.PixelNeRF):
    """Stands in for the HIP renderer on CPU: per-ray outputs that depend on the ray, on its position inside its
    reference chunk (as the real direction-tiling quirk does) and on the whole-tensor keys (src_poses)."""

    def forward(self, rays, randomized, white_bkgd, near, far, chunk=None):
        o = rays["rays_o"]
        B = o.shape[0]
        chunk = int(chunk or max(B, 1))
        pos = (torch.arange(B) % chunk).float()[:, None]                      # position inside the reference chunk
        bias = rays["src_poses"].sum()                                        # whole-tensor key must arrive unsliced
        rgb = o * 2.0 + pos * 1e-3 + bias
        depth = o.sum(-1) + float(near)
        acc = o[:, 0] * 0.5 + float(far)
        return [(rgb * 0, acc * 0, depth * 0), (rgb, acc, depth)]


def _sharded_worker(rank, world, port, n, chunk, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from neo360_amd import render
        g = torch.Generator().manual_seed(11)
        batch = dict(rays_o=torch.randn(n, 3, generator=g), rays_d=torch.randn(n, 3, generator=g),
                     viewdirs=torch.randn(n, 3, generator=g), src_poses=torch.randn(3, 4, 4, generator=g),
                     src_focal=torch.ones(3), src_c=torch.zeros(3, 2), src_imgs=torch.zeros(3, 3, 4, 4))
        net = _FakePixelNeRF()
        whole = render.render_rays_test(net, batch, chunk=chunk, near=0.2, far=2.5)
        want = torch.cat([whole["rgb"], whole["depth"][:, None], whole["acc"][:, None]], dim=1)
        got = render.render_frame_sharded(net, batch, world, rank, chunk=chunk, near=0.2, far=2.5)
        ret[rank] = bool(torch.equal(got, want))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,world,chunk", [(1500, 2, 256), (1000, 3, 128)])
def test_sharded_frame_equals_whole_frame_gloo(n, world, chunk):
    """Shards are whole reference chunks, per-ray keys are sliced, src_* keys are passed whole, and the gathered
    (rgb, depth, acc) frame equals the single-process frame bit for bit — on every rank."""
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_sharded_worker, args=(world, port, n, chunk, ret), nprocs=world, join=True)
        assert all(ret.get(r) for r in range(world)), dict(ret)


def _world1_worker(rank, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from neo360_amd import render
        g = torch.Generator().manual_seed(5)
        n = 700
        batch = dict(rays_o=torch.randn(n, 3, generator=g), rays_d=torch.randn(n, 3, generator=g),
                     viewdirs=torch.randn(n, 3, generator=g), src_poses=torch.randn(3, 4, 4, generator=g),
                     src_focal=torch.ones(3), src_c=torch.zeros(3, 2), src_imgs=torch.zeros(3, 3, 4, 4))
        net = _FakePixelNeRF()
        plain = render.render_frame_sharded(net, batch, 1, 0, chunk=256)                       # no collective at world 1
        coll = render.render_frame_sharded(net, batch, 1, 0, chunk=256, always_gather=True)   # one-rank all-gather
        ret[0] = bool(torch.equal(plain, coll)) and coll.data_ptr() != plain.data_ptr()
    finally:
        dist.destroy_process_group()


def test_always_gather_runs_the_collective_at_world_one():
    """bench.py under `torchrun --nproc-per-node 1` (tests/test_gpu_multirank.py runs it on RCCL): same frame."""
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_world1_worker, args=(port, ret), nprocs=1, join=True)
        assert ret.get(0) is True


def _agc_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sizes = [(4, 6), (3, 5)]
        n = sum(h * w for h, w in sizes)
        assert n % world == 0
        rows = n // world                                  # every rank holds the same number of rows (Lightning's all_gather)
        outs_rgb = [{"rgb": torch.arange(rows * 3, dtype=torch.float32).reshape(rows, 3) + 1000.0 * rank}]
        outs_d = [{"depth": torch.arange(rows, dtype=torch.float32).reshape(rows, 1) + 1000.0 * rank}]
        got_rgb = parallel.alter_gather_cat(outs_rgb, "rgb", sizes)
        got_d = parallel.alter_gather_cat(outs_d, "depth", sizes)
        # the reference's arithmetic restated on the stacked per-rank tensors (interface.py:31-36: all_gather -> (world, n, C) ->
        # permute(1, 0, 2).flatten(0, 1) -> squeeze a trailing 1)
        every_rgb = torch.stack([torch.arange(rows * 3, dtype=torch.float32).reshape(rows, 3) + 1000.0 * r for r in range(world)])
        every_d = torch.stack([torch.arange(rows, dtype=torch.float32).reshape(rows, 1) + 1000.0 * r for r in range(world)])
        want_rgb = every_rgb.permute(1, 0, 2).flatten(0, 1)
        want_d = every_d.permute(1, 0, 2).flatten(0, 1).squeeze(-1)
        ok, curr = True, 0
        for i, (h, w) in enumerate(sizes):
            ok = ok and bool(torch.equal(got_rgb[i], want_rgb[curr:curr + h * w].reshape(h, w, 3)))
            ok = ok and bool(torch.equal(got_d[i], want_d[curr:curr + h * w].reshape(h, w)))
            curr += h * w
        ret[rank] = ok and len(got_rgb) == 2 and got_rgb[0].shape == (4, 6, 3) and got_d[1].shape == (3, 5)
    finally:
        dist.destroy_process_group()


def test_alter_gather_cat_world_3_matches_the_reference_reordering():
    """models/interface.py:30-50 for a list of images per rank: one all_gather_into_tensor + the reference's row interleave."""
    world, port = 3, _free_port()                          # 24 + 15 = 39 rows = 3 x 13
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_agc_worker, args=(world, port, ret), nprocs=world, join=True)
        assert all(ret.get(r) is True for r in range(world)), dict(ret)


def test_alter_gather_cat_single_process_is_the_identity_gather():
    sizes = [(2, 3), (1, 4)]
    outs = [{"rgb": torch.arange(18, dtype=torch.float32).reshape(6, 3)}, {"rgb": torch.arange(12, dtype=torch.float32).reshape(4, 3) + 50.0}]
    got = parallel.alter_gather_cat(outs, "rgb", sizes)
    assert torch.equal(got[0], outs[0]["rgb"].reshape(2, 3, 3)) and torch.equal(got[1], outs[1]["rgb"].reshape(1, 4, 3))
    d = parallel.alter_gather_cat([{"depth": torch.arange(10, dtype=torch.float32)}], "depth", sizes)
    assert torch.equal(d[0], torch.arange(6, dtype=torch.float32).reshape(2, 3)) and d[1].shape == (1, 4)


def test_ray_patch_order_is_a_permutation_for_any_shard():
    """The evaluators' ray-patch tile order (neo_ctx_set_ray_grid; host mirror parallel.ray_patch_order of tp_common.h:patch_point)
    permutes the rays of a launch - whole frame, band-aligned shard, shard with ragged ends, any patch shape."""
    import torch
    from neo360_amd.parallel import ray_patch_order, shard_bounds
    W, H = 640, 480
    for pw, ph in ((1, 1), (3, 3), (2, 2), (0, 3), (1, 3)):
        o = ray_patch_order(W * H, W, 0, pw, ph)
        assert torch.equal(torch.sort(o).values, torch.arange(W * H))
        # first patch of the frame: 2^pw x 2^ph pixels, row-major
        want = torch.tensor([y * W + x for y in range(1 << ph) for x in range(1 << pw)])
        assert torch.equal(o[:want.numel()], want)
    for world in (2, 3, 8):
        for rank in range(world):
            lo, hi = shard_bounds(W * H, world, rank, unit=1024)
            o = ray_patch_order(hi - lo, W, lo, 1, 1)
            assert torch.equal(torch.sort(o).values, torch.arange(hi - lo)), (world, rank)
            # rays outside whole bands of the shard keep their place
            band = W << 1
            head = (-lo) % band
            assert torch.equal(o[:head], torch.arange(head))
    assert torch.equal(ray_patch_order(1000, 0), torch.arange(1000))          # no hint
    assert torch.equal(ray_patch_order(1000, 60, 0, 3, 3), torch.arange(1000))   # width not a multiple of the patch width
