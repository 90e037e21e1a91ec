# round 6: quick look after a training-kernel change: the chain tests, two step timings, kernel stats of the step
cd $GRAFT_REPO_ROOT; T=${1:-r06t}; O=gpurun_out/$T; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_host_r6.py -q -m gpu -x -k "fused_training_chain or merged_projection" > $O/pytest_a.log 2>&1; echo "pytest rc=$?" >> $O/pytest_a.log
tail -3 $O/pytest_a.log
grep -q "rc=0" $O/pytest_a.log || exit 1
for m in 1 1; do
  timeout 150 python bench.py --workload neo360_train --steps 10 --warmup 3 --cpu-rays 0 > $O/train_$m.json 2> $O/train_$m.err || { tail -3 $O/train_$m.err; continue; }
  python - <<PY
import json
t = json.load(open("$O/train_$m.json")); print("train: ms/step %.2f rays/s %.0f" % (t["ms_per_step"], t["value"]), t.get("phases_ms"))
PY
done
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_train -o train -- python bench.py --workload neo360_train --steps 5 --warmup 2 --cpu-rays 0 > $O/prof_train.log 2>&1
find $O/prof_train -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_train.csv \;
find $O/prof_train -name "*kernel_trace.csv" -delete; find $O/prof_train -name "*agent_info.csv" -delete
head -30 $O/kernel_stats_train.csv | cut -c1-110,200-260
