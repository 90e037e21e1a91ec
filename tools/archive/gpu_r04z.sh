# round 4's closing GPU call: full GPU test suite, smoke, counter passes of the bench command (both evaluator kernels), the
# driver's bench command, rocprofv3 kernel stats of the same workload, the kernel micro-benchmark of the four launches.
# usage: bash tools/gpu_r04z.sh [tag]
cd $GRAFT_REPO_ROOT; T=${1:-r04z}; O=gpurun_out/$T; rm -rf $O; mkdir -p $O
rm -f gpurun_out/parity_report.json
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --durations=10 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
# counters first: the bench line below takes its traffic / mfma_busy fields from the summary of THIS tree (stamped with the hash of
# the kernels' sources; bench.py drops them when the stamp does not match)
bash tools/pmc_bench.sh neo360 f16x3 "k_tp_mlp_hp<" "k_tp_mlp_hpp<" > $O/pmc_bench.log 2>&1
cp gpurun_out/pmc_neo360_f16x3/summary.json profiles/r04_pmc_neo360_f16x3.json
cp gpurun_out/pmc_neo360_f16x3/summary_2.json profiles/r04_pmc_neo360_f16x3_hpp.json
cp gpurun_out/pmc_neo360_f16x3/kernel_stats.csv $O/kernel_stats_pmc_run.csv
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- python bench.py --steps 5 --warmup 1 --cpu-rays 0 --others 0 --exact-f32 0 > $O/prof.log 2>&1
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*agent_info.csv" -delete
for pp in 1 2 3; do PP=$pp TAG=pp$pp SLOTS=1,3,0,2 REPS=40 timeout 300 python tools/bench_tp_kernel.py 2>&1 | grep -E "slot" >> $O/kernel_launches.log; done
find gpurun_out -name "*counter_collection.csv" -size +200k -delete
tail -6 $O/pytest.log; tail -2 $O/smoke.log; cut -c1-400 $O/bench.json; cat $O/kernel_launches.log
python - <<PY
import json
d = json.load(open("$O/bench.json")); r = d["roofline"]
print("rays/s %.0f  ms %.1f  kernel %s  avg launch %.2f ms  frac %.3f  frac_executed %s  mfma_busy %s  sclk %s  power %s" % (
    d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"], r["frac"], r.get("frac_executed"), r.get("mfma_busy"), r.get("sclk_mhz_mean"), r.get("power_w_mean")))
print({k: (round(v["avg_launch_ms"], 2), v["launches"], round(v["algorithmic_tflops"], 1)) for k, v in r.get("kernels", {}).items()})
print("exact_f32", d.get("exact_f32")); print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("speedup_vs_cpu")); print({k: round(v["value"]) for k, v in d.get("other_workloads", {}).items()})
PY
