"""Per-wave barrier timeline of ONE workgroup of k_tp_mlp_hp (variant built with -DNEO_TP_TIMELINE=1): s_memtime before and
after every barrier for each of the 4 waves.  Prints, per barrier interval: the time each wave spent working (end of the
previous barrier -> arrival at this one) and waiting (arrival -> release), aggregated per phase of the view loop.
env: SLOT, BLOCK (workgroup to trace, default 1000)."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("REPS", "1")
BLOCK = int(os.environ.get("BLOCK", 1000))
from neo360_amd import _lib
lib = _lib.load()
buf = (ctypes.c_ulonglong * 4096)()
lib.neo_debug_tp_stamps(buf, BLOCK)
exec(open(os.path.join(ROOT, "tools", "bench_tp_kernel.py")).read().split("if os.environ.get(\"TRACE\")")[0])
lib.neo_debug_tp_stamps(buf, BLOCK)          # the warm-up + timed launches above all stamped the same workgroup: last one wins
st = np.array(buf, dtype=np.uint64).reshape(4, 1024).astype(np.int64)
n = int((st[0] > 0).sum())
print("stamps per wave:", n, "-> barriers:", (n - 2) // 2)
t0 = st[:, 0].min()
st = st[:, :n] - t0
# stamp 0 = kernel start, then (arrive_k, release_k) pairs, last = end
arr, rel = st[:, 1:n - 1:2], st[:, 2:n - 1:2]
prev = np.concatenate([st[:, :1], rel[:, :-1]], axis=1)
work = arr - prev                     # per wave, per barrier: work before arriving
wait = rel - arr                      # time spent in the barrier
print("total %d cycles; per wave work %s  wait %s" % (st[:, -1].max(), work.sum(1).tolist(), wait.sum(1).tolist()))
nb = arr.shape[1]
print("barrier  work(max over waves)  work(min)  wait(max)  wait(min)   release time")
for k in range(nb):
    print("%4d %10d %10d %10d %10d %12d" % (k, work[:, k].max(), work[:, k].min(), wait[:, k].max(), wait[:, k].min(), rel[:, k].max()))
