# round 5: lookup backward with run merging (k_gather_bwd_runs) - gradient tests + the training-step line, A/B against one atomic per (row, tap, channel)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05f; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_host_r4.py -q -m gpu --durations=3 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
for v in runs norun runs2; do
  L=""; [ $v = norun ] && L=$PWD/tools/build/libneo_gather_norun.so
  NEO360_HIP_LIB=$L timeout 600 python bench.py --workload neo360_train --steps 8 --warmup 2 --cpu-rays 0 > $O/train_$v.json 2> $O/train_$v.err
  python -c "
import json; d=json.load(open('$O/train_$v.json')); print('$v: train step ms %.1f' % d['ms_per_step'], {k: round(x, 1) for k, x in d['phases_ms'].items()}, 'rays/s %.0f' % d['value'])" | tee -a $O/train_ab.log
done
