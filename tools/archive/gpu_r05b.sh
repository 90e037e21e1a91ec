# round 5, second GPU call: the new full-size fixtures b5 / b6 + the range-guard latch, then FULL-FRAME A/B of the evaluator
# variants (the micro-benchmark of gpu_r05a.sh is too short to reach the power-limited steady state the frame runs in):
#   default = round-4 kernels   pair2 = doubling-order encodings   mixlo = 3-instruction split   pair2mix = both   all = + chained blends, bias re-read
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05b; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_host_r5.py tests/test_gpu_fullsize.py tests/test_gpu_range_guard.py tests/test_gpu_host_r4.py -q -m gpu --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
B=$PWD/tools/build
frame() { # tag lib
  NEO360_HIP_LIB=$2 timeout 300 python bench.py --steps 10 --warmup 2 --cpu-rays 0 --others 0 --exact-f32 0 --setup-timing 0 > $O/frame_$1.json 2> $O/frame_$1.err
  python - "$1" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.load(open("gpurun_out/r05b/frame_%s.json" % tag)); r = d["roofline"]
    k = {n: round(v["avg_launch_ms"], 2) for n, v in r["kernels"].items()}
    j = {n: round(v["avg_launch_ms"] * 1e-3 * r["power_w_mean"], 1) for n, v in r["kernels"].items()}
    print("%-10s rays/s %8.0f  ms %.1f  %s  sclk %.0f  power %.0f W  joules/launch %s" % (tag, d["value"], d["ms_per_step"], k, r["sclk_mhz_mean"], r["power_w_mean"], j))
except Exception as e:
    print(tag, "failed", e)
PY
}
frame default ""          | tee -a $O/frame_ab.log
frame pair2 $B/libneo_pair2.so       | tee -a $O/frame_ab.log
frame mixlo $B/libneo_mixlo.so       | tee -a $O/frame_ab.log
frame pair2mix $B/libneo_pair2mix.so | tee -a $O/frame_ab.log
frame all $B/libneo_all.so           | tee -a $O/frame_ab.log
frame default2 ""         | tee -a $O/frame_ab.log
frame pair2b $B/libneo_pair2.so      | tee -a $O/frame_ab.log
