# round 5: run-merged scatter kernels - channel order of a lane group (one 64-byte sector per atomic instruction) and a scope probe
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05o; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_host_r5.py -q -m gpu -k "gather or end_to_end or projected or differentiable" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for v in new sc_old sc_wg new2; do
  L=""; [ $v = sc_old -o $v = sc_wg ] && L=$PWD/tools/build/libneo_$v.so
  NEO360_HIP_LIB=$L timeout 600 python bench.py --workload neo360_train --steps 8 --warmup 2 --cpu-rays 0 > $O/train_$v.json 2> $O/train_$v.err
  python -c "
import json; d=json.load(open('$O/train_$v.json')); print('$v: train step ms %.1f' % d['ms_per_step'], {k: round(x, 1) for k, x in d['phases_ms'].items()}, 'rays/s %.0f' % d['value'])" | tee -a $O/train_ab.log
done
