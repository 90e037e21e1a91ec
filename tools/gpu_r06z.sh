# round 6's closing GPU call: full GPU test suite, smoke, counter passes of the bench command for the NeO-360 evaluators, the vanilla and
# the Mip-NeRF 360 kernels (stamped with the hash of the sources they ran on), the driver's bench command, rocprofv3 kernel stats of the
# same workload, the training-step line + its kernel stats.
# usage: bash tools/gpu_r06z.sh [tag]
cd $GRAFT_REPO_ROOT; T=${1:-r06z}; O=gpurun_out/$T; rm -rf $O; mkdir -p $O
rm -f gpurun_out/parity_report.json
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
bash tools/pmc_bench.sh neo360 f16x3 "k_tp_mlp_hp<" "k_tp_mlp_hpp<" > $O/pmc_bench.log 2>&1
cp gpurun_out/pmc_neo360_f16x3/summary.json profiles/r06_pmc_neo360_f16x3.json
cp gpurun_out/pmc_neo360_f16x3/summary_2.json profiles/r06_pmc_neo360_f16x3_hpp.json
cp gpurun_out/pmc_neo360_f16x3/kernel_stats.csv $O/kernel_stats_pmc_run.csv
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- python bench.py --steps 5 --warmup 1 --cpu-rays 0 --others 0 --exact-f32 0 --chunk-loop 0 > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_neo360.csv \;
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*agent_info.csv" -delete
bash tools/pmc_bench.sh vanilla f16x3 "k_vanilla_mlp_h" > $O/pmc_bench_vanilla.log 2>&1
cp gpurun_out/pmc_vanilla_f16x3/summary.json profiles/r06_pmc_vanilla_f16x3.json
bash tools/pmc_bench.sh mip360 f16x3 "k_mip_gemm_h" "k_mip_mlp_h<1024" > $O/pmc_bench_mip.log 2>&1
cp gpurun_out/pmc_mip360_f16x3/summary.json profiles/r06_pmc_mip360_f16x3.json
cp gpurun_out/pmc_mip360_f16x3/kernel_stats.csv $O/kernel_stats_mip360.csv
python bench.py --workload neo360_train --steps 10 --warmup 2 > $O/train.json 2> $O/train.err
rocprofv3 --kernel-trace --stats -f csv -d $O/prof_train -o train -- python bench.py --workload neo360_train --steps 5 --warmup 2 --cpu-rays 0 > $O/prof_train.log 2>&1
find $O/prof_train -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_train.csv \;
find $O/prof_train -name "*kernel_trace.csv" -delete; find $O/prof_train -name "*agent_info.csv" -delete
cp gpurun_out/parity_report.json $O/parity_report.json; timeout 300 python tools/bench_train_other.py > $O/train_other.log 2>&1; timeout 200 python tools/bench_train_chain.py > $O/train_chain.log 2>&1; NEO360_TRAIN_CHAIN=0 timeout 200 python tools/bench_train_chain.py >> $O/train_chain.log 2>&1
find gpurun_out -name "*counter_collection.csv" -size +200k -delete
find gpurun_out -name "*kernel_trace.csv" -size +200k -delete
tail -6 $O/pytest.log; tail -2 $O/smoke.log; cut -c1-300 $O/bench.json
python - <<PY
import json
d = json.load(open("$O/bench.json")); r = d["roofline"]
print("rays/s %.0f  ms %.1f  kernel %s  avg launch %.2f ms  frac %.3f  frac_executed %s  mfma_busy %s  traffic %s  sclk %s  power %s" % (
    d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"], r["frac"], r.get("frac_executed"), r.get("mfma_busy"), r.get("traffic"), r.get("sclk_mhz_mean"), r.get("power_w_mean")))
print({k: (round(v["avg_launch_ms"], 2), v["launches"], round(v["algorithmic_tflops"], 1)) for k, v in r.get("kernels", {}).items()})
print("scene_setup", d.get("scene_setup", {}).get("runs_ms"), d.get("scene_setup_ms"))
print("chunk_loop", d.get("chunk_loop", {}).get("value"), d.get("chunk_loop", {}).get("frac_of_headline"), "energy/step", r.get("energy_j_per_step")); print("exact_f32", d.get("exact_f32")); print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("speedup_vs_cpu")); print({k: (round(v["value"]), (v.get("chunk_loop") or {}).get("value")) for k, v in d.get("other_workloads", {}).items()})
t = json.load(open("$O/train.json")); print("train: ms/step %.1f rays/s %.0f" % (t["ms_per_step"], t["value"]), t["phases_ms"], "cpu", t.get("cpu_baseline", {}).get("value"))
PY
