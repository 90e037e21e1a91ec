# round 4, second GPU call: the work-list gather of mlp_tp_hpp.hip vs the whole-map-variant version and the round-3 kernel;
# per-slot mode 3; the exact-fp32 evaluator on projected maps; the new tests (differentiable module call, fp32 projected
# stages, full-size chunks b1-b4 with the tightened rule).   usage: bash tools/gpu_r04b.sh
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04b; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_neo360_stages.py tests/test_gpu_neo360.py tests/test_gpu_training.py tests/test_gpu_host_r4.py \
   tests/test_gpu_repeatable.py tests/test_gpu_range_guard.py -q -m gpu -x --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -20 $O/pytest.log
K=$O/kernel_ab.log
run() { NEO360_HIP_LIB=$2 PP=$3 TAG=$1 SLOTS=1,3,0,2 REPS=40 timeout 300 python tools/bench_tp_kernel.py 2>&1 | grep -E "slot|Error|error" >> $K; }
B=$PWD/tools/build
run hp        "" 1
run hpp_wl    "" 2
run hpp_stat  $B/libneo_hpp_static.so 2
run hpp_st1   $B/libneo_hpp_st1.so 2
run hpp_xd2   $B/libneo_hpp_wps3.so 2
run hp        "" 1
run hpp_wl    "" 2
cat $K
for pp in 1 2 3; do NEO360_TP_PREPROJECT=$pp timeout 600 python bench.py --steps 8 --warmup 2 --cpu-rays 0 --others 0 --exact-f32 0 > $O/bench_pp$pp.json 2> $O/bench_pp$pp.err; tail -2 $O/bench_pp$pp.err; done
for pp in 1 2; do NEO360_TP_PREPROJECT=$pp timeout 600 python bench.py --precision f32 --steps 2 --warmup 1 --cpu-rays 0 --others 0 --exact-f32 0 > $O/bench_f32_pp$pp.json 2> $O/bench_f32_pp$pp.err; tail -2 $O/bench_f32_pp$pp.err; done
python - <<'PY'
import json
for name in ("pp1", "pp2", "pp3", "f32_pp1", "f32_pp2"):
    try:
        d = json.load(open("gpurun_out/r04b/bench_%s.json" % name))
        r = d["roofline"]
        print(name, "rays/s %.0f" % d["value"], "ms %.1f" % d["ms_per_step"], "kernel", r["kernel"], "avg launch ms %.2f" % r["avg_launch_ms"],
              "sclk %.0f" % (r.get("sclk_mhz_mean") or 0), "power %.0f" % (r.get("power_w_mean") or 0), "frac %.3f" % r["frac"], "setup ms", d.get("scene_setup_ms"))
    except Exception as e:
        print(name, "failed", e)
PY
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -m gpu --durations=5 > $O/pytest_full.log 2>&1; echo "pytest rc=$?" >> $O/pytest_full.log
tail -12 $O/pytest_full.log
