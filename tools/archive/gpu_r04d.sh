# round 4: the outside-sphere evaluator's two small savings (empty tile-views skip the gather; 1/r encodings cached across views)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04d; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_neo360_stages.py tests/test_gpu_neo360.py tests/test_gpu_repeatable.py tests/test_gpu_fullsize.py tests/test_gpu_multirank.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
K=$O/kernel_ab.log
run() { NEO360_HIP_LIB=$2 PP=$3 TAG=$1 SLOTS=3,2 REPS=60 timeout 300 python tools/bench_tp_kernel.py 2>&1 | grep -E "slot|Error|error" >> $K; }
B=$PWD/tools/build
run hpp_prev $B/libneo_hpp_prev.so 2
run hpp_se   $B/libneo_hpp_se.so 2
run hpp_wc   $B/libneo_hpp_wc.so 2
run hpp_both "" 2
run hpp_prev $B/libneo_hpp_prev.so 2
run hpp_both "" 2
cat $K
timeout 600 python bench.py --steps 8 --warmup 2 --cpu-rays 0 --others 0 --exact-f32 0 > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04d/bench.json")); r = d["roofline"]
print("rays/s %.0f ms %.1f" % (d["value"], d["ms_per_step"]), {k: round(v["avg_launch_ms"], 2) for k, v in r["kernels"].items()}, r.get("sclk_mhz_mean"), r.get("power_w_mean"), r.get("pmc_stale"))
PY
