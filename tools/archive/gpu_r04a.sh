# round 4, first GPU call: correctness of the new host paths + the pre-projected-planes evaluator, then the kernel A/B:
#   hp_r3   = round-3 kernel (no descriptor prefetch, 8-byte epilogue stores)
#   hp_dpf  = + tap descriptors read from LDS one item ahead        hp_st = + 16-byte permlane-swapped epilogue stores
#   default = both                                                   PP=2  = mlp_tp_hpp.hip (tri-planes pre-projected too)
# usage: bash tools/gpu_r04a.sh
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04a; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_host_r4.py tests/test_gpu_multirank.py tests/test_gpu_neo360.py tests/test_gpu_neo360_stages.py \
   tests/test_gpu_repeatable.py tests/test_gpu_range_guard.py tests/test_gpu_flags.py -q -m gpu -x --durations=10 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
K=$O/kernel_ab.log
run() { # tag lib pp
  NEO360_HIP_LIB=$2 PP=$3 TAG=$1 SLOTS=1,3,0,2 REPS=40 timeout 300 python tools/bench_tp_kernel.py 2>&1 | grep -E "slot|Error|error" >> $K; }
B=$PWD/tools/build
run hp_r3   $B/libneo_hp_r3.so 1
run hp_dpf  $B/libneo_hp_dpf.so 1
run hp_st   $B/libneo_hp_st.so 1
run hp_both "" 1
run hpp     "" 2
run hpp_m0  $B/libneo_hpp_m0.so 2
run hpp_z0  $B/libneo_hpp_z0.so 2
run hpp_xd1 $B/libneo_hpp_xd1.so 2
run hpp_r4  $B/libneo_hpp_r4.so 2
run hp_r3   $B/libneo_hp_r3.so 1
run hp_both "" 1
run hpp     "" 2
cat $K
# whole frame, both modes, with the telemetry fields
for pp in 1 2; do NEO360_TP_PREPROJECT=$pp timeout 600 python bench.py --steps 8 --warmup 2 --cpu-rays 0 --others 0 --exact-f32 0 > $O/bench_pp$pp.json 2> $O/bench_pp$pp.err; cut -c1-400 $O/bench_pp$pp.json; tail -3 $O/bench_pp$pp.err; done
python - <<'PY'
import json
for pp in (1, 2):
    try:
        d = json.load(open("gpurun_out/r04a/bench_pp%d.json" % pp))
        r = d["roofline"]
        print("pp", pp, "rays/s %.0f" % d["value"], "ms %.1f" % d["ms_per_step"], "kernel", r["kernel"], "avg launch ms %.2f" % r["avg_launch_ms"],
              "sclk", r.get("sclk_mhz_mean"), "power", r.get("power_w_mean"), "limit", r.get("power_limit_w"), "samples", r.get("telemetry_samples"),
              "frac", r["frac"], "frac@clk", r.get("frac_at_measured_clock"))
    except Exception as e:
        print("pp", pp, "failed", e)
PY
