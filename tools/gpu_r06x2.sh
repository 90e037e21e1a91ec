# round 6: XCD-aware workgroup -> tile map of the training GEMMs (k_sgemm column tiles of a row block / k_dw tiles of a K slice on one XCD):
# tests, A/B of the step lines against a NEO_GEMM_XCD=0 build, HBM traffic of the step with the new map
cd $GRAFT_REPO_ROOT; T=${1:-r06x2}; O=gpurun_out/$T; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python tools/build_variant.py noxcd train_mlp.hip -DNEO_GEMM_XCD=0 > /dev/null 2>&1
timeout 400 python -m pytest tests/test_gpu_host_r6.py tests/test_gpu_training.py tests/test_gpu_host_r5.py tests/test_gpu_pix_training.py tests/test_gpu_mip_training.py -q -m gpu -x > $O/pytest_a.log 2>&1; echo "pytest rc=$?" >> $O/pytest_a.log
tail -3 $O/pytest_a.log
grep -q "rc=0" $O/pytest_a.log || exit 1
for v in xcd plain xcd plain; do
  if [ $v = plain ]; then export NEO360_HIP_LIB=tools/build/libneo_noxcd.so; else unset NEO360_HIP_LIB; fi
  timeout 150 python bench.py --workload neo360_train --steps 10 --warmup 3 --cpu-rays 0 > $O/train_$v.json 2> $O/train_$v.err || { tail -3 $O/train_$v.err; continue; }
  python - <<PY
import json
t = json.load(open("$O/train_$v.json")); print("$v train: ms/step %.2f rays/s %.0f" % (t["ms_per_step"], t["value"]), t.get("phases_ms"))
PY
done
for v in xcd plain; do
  if [ $v = plain ]; then export NEO360_HIP_LIB=tools/build/libneo_noxcd.so; else unset NEO360_HIP_LIB; fi
  echo "== $v"; timeout 200 python tools/bench_train_other.py 2>&1 | grep -v amdgpu | tee $O/train_other_$v.log
done
unset NEO360_HIP_LIB
CMD="python $GRAFT_REPO_ROOT/bench.py --workload neo360_train --steps 3 --warmup 1 --cpu-rays 0"
i=0
for C in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C -f csv -d $GRAFT_REPO_ROOT/$O/pass$i -- $CMD > $GRAFT_REPO_ROOT/$O/pass$i.log 2>&1)
done
python tools/pmc_train_step.py $O 6 > $O/pmc_train_step.json; head -24 $O/pmc_train_step.json
find $O -name "*.csv" -size +1M -delete; find $O -name "*agent_info.csv" -delete
