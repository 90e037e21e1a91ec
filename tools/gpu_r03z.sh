# the round's closing GPU call: full GPU test suite, smoke, the default bench line (driver arguments), rocprofv3 kernel stats of
# the same workload, the counter passes (bench command + per-kernel micro-benchmark).  usage: bash tools/gpu_r03z.sh
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03z; rm -rf $O; mkdir -p $O
rm -f gpurun_out/parity_report.json
python -m pytest tests -q -m gpu --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- python bench.py --steps 5 --warmup 1 --cpu-rays 0 --others 0 --exact-f32 0 > $O/prof.log 2>&1
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*agent_info.csv" -delete
bash tools/pmc_bench.sh neo360 f16x3 k_tp_mlp_hp > $O/pmc_bench.log 2>&1
bash tools/pmc_tp2.sh r03z/pmc_fg SLOT=1 > $O/pmc_fg.log 2>&1
bash tools/pmc_tp2.sh r03z/pmc_bg SLOT=3 > $O/pmc_bg.log 2>&1
find gpurun_out -name "*counter_collection.csv" -size +200k -delete
tail -4 $O/pytest.log; tail -2 $O/smoke.log; cut -c1-300 $O/bench.json
