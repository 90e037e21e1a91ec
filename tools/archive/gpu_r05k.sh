# round 5: what a launch costs per MATRIX instruction - the evaluators with one / both cross terms of every split product dropped
# (NEO_SPLIT_TERMS=2 / 1: wrong results, timing + energy only).  Bounds what a cheaper cross-term arithmetic could buy.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05k; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
B=$PWD/tools/build
frame() { # tag lib
  NEO360_HIP_LIB=$2 timeout 300 python bench.py --steps 10 --warmup 2 --cpu-rays 0 --others 0 --exact-f32 0 --setup-timing 0 > $O/frame_$1.json 2> $O/frame_$1.err
  python - "$1" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.load(open("gpurun_out/r05k/frame_%s.json" % tag)); r = d["roofline"]
    k = {n: round(v["avg_launch_ms"], 2) for n, v in r["kernels"].items()}
    j = {n: round(v["avg_launch_ms"] * 1e-3 * r["power_w_mean"], 1) for n, v in r["kernels"].items()}
    print("%-10s rays/s %8.0f  ms %.1f  %s  sclk %.0f  power %.0f W  joules/launch %s" % (tag, d["value"], d["ms_per_step"], k, r["sclk_mhz_mean"], r["power_w_mean"], j))
except Exception as e:
    print(tag, "failed", e)
PY
}
frame default ""                | tee -a $O/frame_ab.log
frame terms2 $B/libneo_terms2.so | tee -a $O/frame_ab.log
frame terms1 $B/libneo_terms1.so | tee -a $O/frame_ab.log
frame default2 ""               | tee -a $O/frame_ab.log
frame terms2b $B/libneo_terms2.so | tee -a $O/frame_ab.log
