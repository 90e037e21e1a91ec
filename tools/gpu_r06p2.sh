# round 6: training tests, then HBM traffic of the training step (two counter-only passes) -> profiles/r06_pmc_train_step.json (stamped)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06p2; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_host_r6.py tests/test_gpu_training.py tests/test_gpu_host_r5.py tests/test_gpu_pix_training.py tests/test_gpu_mip_training.py tests/test_gpu_pixelnerf.py -q -m gpu -x > $O/pytest_a.log 2>&1; echo "pytest rc=$?" >> $O/pytest_a.log
tail -3 $O/pytest_a.log
CMD="python $GRAFT_REPO_ROOT/bench.py --workload neo360_train --steps 3 --warmup 1 --cpu-rays 0"
i=0
for C in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C -f csv -d $GRAFT_REPO_ROOT/$O/pass$i -- $CMD > $GRAFT_REPO_ROOT/$O/pass$i.log 2>&1)
done
python tools/pmc_train_step.py $O 6 > $O/pmc_train_step.json; head -12 $O/pmc_train_step.json
find $O -name "*.csv" -size +1M -delete; find $O -name "*agent_info.csv" -delete
timeout 150 python bench.py --workload neo360_train --steps 10 --warmup 3 --cpu-rays 0 > $O/train.json 2> $O/train.err
python - <<PY
import json
t = json.load(open("$O/train.json")); print("train: ms/step %.2f" % t["ms_per_step"], t["roofline"].get("traffic"), t["roofline"].get("hbm_frac"))
PY
