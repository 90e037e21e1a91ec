# round 5: exact-fp32 NeO-360 evaluator (k_tp_mlp) with the register budget of three workgroups per CU, frame A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05x; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
B=$PWD/tools/build
frame() { # tag lib
  NEO360_HIP_LIB=$2 timeout 300 python bench.py --precision f32 --steps 3 --warmup 1 --cpu-rays 0 --others 0 --exact-f32 0 --setup-timing 0 > $O/frame_$1.json 2> $O/frame_$1.err
  python - "$1" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.load(open("gpurun_out/r05x/frame_%s.json" % tag)); r = d["roofline"]
    k = {n: round(v["avg_launch_ms"], 2) for n, v in r["kernels"].items()}
    print("%-10s rays/s %8.0f  ms %.1f  %s  sclk %.0f  power %.0f W  frac_executed %s" % (tag, d["value"], d["ms_per_step"], k, r["sclk_mhz_mean"], r["power_w_mean"], r.get("frac_executed")))
except Exception as e:
    print(tag, "failed", e)
PY
}
frame wg2 ""                 | tee -a $O/frame_ab.log
frame wg3 $B/libneo_f32wg3.so | tee -a $O/frame_ab.log
frame wg2b ""                | tee -a $O/frame_ab.log
