# round 6: HBM traffic of the training step (two counter-only passes) + the driver's default line once more (scene_setup fix)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06p2; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --workload neo360_train --steps 3 --warmup 1 --cpu-rays 0"
i=0
for C in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C -f csv -d $GRAFT_REPO_ROOT/$O/pass$i -- $CMD > $GRAFT_REPO_ROOT/$O/pass$i.log 2>&1)
done
python tools/pmc_train_step.py $O 6 > $O/pmc_train_step.json; cat $O/pmc_train_step.json
find $O -name "*.csv" -size +1M -delete; find $O -name "*agent_info.csv" -delete
timeout 400 python bench.py --gpus 1 --steps 5 --warmup 2 --cpu-rays 0 --others 0 --exact-f32 0 --chunk-loop 0 > $O/bench_short.json 2> $O/bench_short.err
python - <<PY
import json
d = json.load(open("$O/bench_short.json")); print(d["value"], d["scene_setup"]["runs_ms"], d["scene_setup_ms"])
PY
