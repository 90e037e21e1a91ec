# the round's closing GPU call: full GPU test suite, smoke, the default bench line (driver arguments), rocprofv3 kernel stats of
# the same workload, the counter passes (bench command + per-kernel micro-benchmark), the phase trace of the final kernel
# (variant tools/build/libneo_trace.so = -DNEO_TP_TRACE=1), the training-MLP throughput.  usage: bash tools/gpu_r03z.sh
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03z; rm -rf $O; mkdir -p $O
rm -f gpurun_out/parity_report.json
python -m pytest tests -q -m gpu --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
export TMPDIR=/tmp
# counters first: the bench line below takes its traffic / mfma_busy fields from the summary of THIS tree (stamped with the hash of
# the kernel's sources; bench.py drops them when the stamp does not match)
bash tools/pmc_bench.sh neo360 f16x3 k_tp_mlp_hp > $O/pmc_bench.log 2>&1
cp gpurun_out/pmc_neo360_f16x3/summary.json profiles/r03_pmc_neo360_f16x3.json
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- python bench.py --steps 5 --warmup 1 --cpu-rays 0 --others 0 --exact-f32 0 > $O/prof.log 2>&1
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*agent_info.csv" -delete
bash tools/pmc_tp2.sh r03z/pmc_fg SLOT=1 > $O/pmc_fg.log 2>&1
bash tools/pmc_tp2.sh r03z/pmc_bg SLOT=3 > $O/pmc_bg.log 2>&1
for s in 1 3; do NEO360_HIP_LIB=$PWD/tools/build/libneo_trace.so TRACE=1 SLOT=$s REPS=3 timeout 200 python tools/bench_tp_kernel.py 2>&1 | tail -12 >> $O/phase_trace.log; done
for s in 1 3 0 2; do N=385; if [ $s = 0 ] || [ $s = 2 ]; then N=129; fi; SLOT=$s N=$N REPS=5 timeout 120 python tools/bench_tp_kernel.py 2>&1 | tail -1 >> $O/kernel_launches.log; done
timeout 300 python tools/bench_train_mlp.py > $O/train_mlp.log 2>&1
find gpurun_out -name "*counter_collection.csv" -size +200k -delete
tail -4 $O/pytest.log; tail -2 $O/smoke.log; cut -c1-300 $O/bench.json; cat $O/phase_trace.log $O/kernel_launches.log $O/train_mlp.log
