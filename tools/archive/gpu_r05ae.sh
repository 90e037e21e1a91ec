# exact-fp32 evaluator with the projected-map gather as a ring of load items: parity of the f32 paths, frames, phase trace
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05ae; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_neo360.py tests/test_gpu_neo360_stages.py tests/test_gpu_fullsize.py tests/test_gpu_range_guard.py tests/test_gpu_repeatable.py -q -m gpu -k "f32 or exact or retry or repeat" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
B=$PWD/tools/build
frame() { # tag lib
  NEO360_HIP_LIB=$2 timeout 300 python bench.py --precision f32 --steps 3 --warmup 1 --cpu-rays 0 --others 0 --exact-f32 0 --setup-timing 0 > $O/frame_$1.json 2> $O/frame_$1.err
  python - "$1" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.load(open("gpurun_out/r05ae/frame_%s.json" % tag)); r = d["roofline"]
    k = {n: round(v["avg_launch_ms"], 2) for n, v in r["kernels"].items()}
    print("%-10s rays/s %8.0f  ms %.1f  %s  sclk %.0f  power %.0f W  frac_executed %s" % (tag, d["value"], d["ms_per_step"], k, r["sclk_mhz_mean"], r["power_w_mean"], r.get("frac_executed")))
except Exception as e:
    print(tag, "failed", e)
PY
}
frame ring4 ""                | tee -a $O/frame_ab.log
frame ring6 $B/libneo_ring6.so | tee -a $O/frame_ab.log
frame ring3 $B/libneo_ring3.so | tee -a $O/frame_ab.log
frame ring4b ""               | tee -a $O/frame_ab.log
NEO360_HIP_LIB=$B/libneo_tp32trace.so PREC=f32 TRACE=f32 SLOT=1 R=8192 REPS=2 TAG=slot1 timeout 300 python tools/bench_tp_kernel.py 2>&1 | grep -v amdgpu | tee -a $O/trace.log
