"""CPU: the launch plumbing of `bench.py --gpus N` (VERDICT r4 task 2): a plain `python bench.py --gpus 2` must start its two
ranks itself (it used to die on `assert WORLD_SIZE == --gpus`), the torch.distributed.run form must keep working, both must
shard the frame by whole 1024-ray chunks and reassemble it in order.  `--fake` swaps the HIP renderer for a CPU stand-in and RCCL
for gloo; nothing is measured."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    return env


def _line(stdout):
    rows = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(rows) == 1, stdout          # rank 0 prints ONE JSON line
    return json.loads(rows[0])


def test_self_spawn_world_2_shards_and_reassembles():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--fake"], env=_env(), cwd="/tmp",
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _line(r.stdout)
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["self_spawned"] is True
    assert out["frame_rows"] == 640 * 480 and out["frame_in_order"] is True and out["rank_column"] == [0.0, 1.0]
    # round 6: one record per rank in the line - "did the collective span N ranks" is answerable from the JSON
    ranks = out["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and all(r["world_size_seen"] == 2 and r["backend"] == "gloo" for r in ranks)
    assert len({r["pid"] for r in ranks}) == 2 and all(r["ms_per_step"] > 0 for r in ranks)
    assert all("numa_node" in r for r in ranks)


def test_torchrun_form_world_2():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--fake"]
    r = subprocess.run(cmd, env=_env(), cwd="/tmp", capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _line(r.stdout)
    assert out["n_gpus"] == 2 and out["self_spawned"] is False and out["frame_in_order"] is True


def test_world_size_mismatch_is_a_message_not_an_assertion():
    env = _env()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29555")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--fake"], env=env, cwd="/tmp", capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "one process per GPU" in r.stderr and "AssertionError" not in r.stderr


def test_more_gpus_than_the_node_has_is_a_clear_message():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "64"], env=_env(), cwd="/tmp", capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 64 but this node shows" in r.stderr and "Traceback" not in r.stderr


def test_numa_binding_is_best_effort():
    sys.path.insert(0, ROOT)
    import bench
    assert bench._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    before = os.sched_getaffinity(0)
    try:
        info = bench.bind_to_gpu_numa_node(0, fake=True)           # no device given: nothing bound, nothing raised
        assert info["numa_node"] is None and info["cpus_bound"] is None
        os.environ["NEO360_FAKE_BDF"] = "ffff:ff:1f.0"              # a PCI address that does not exist
        info = bench.bind_to_gpu_numa_node(0, fake=True)
        assert info["cpus_bound"] is None and "numa_note" in info
    finally:
        os.environ.pop("NEO360_FAKE_BDF", None)
        os.sched_setaffinity(0, before)


def test_single_process_default():
    r = subprocess.run([sys.executable, BENCH, "--fake", "--steps", "1", "--warmup", "0"], env=_env(), cwd="/tmp", capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _line(r.stdout)
    assert out["n_gpus"] == 1 and out["frame_rows"] == 640 * 480
