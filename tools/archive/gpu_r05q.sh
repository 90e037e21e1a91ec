# round 5: k_sgemm with 128 x 128 tiles for N > 64 (forward / dX GEMMs of the training chain), A/B against 128 x 64
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05q; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_host_r5.py -q -m gpu -k "backward or vanilla_training or weight_grad or project or end_to_end" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for v in wide narrow wide2; do
  L=""; [ $v = narrow ] && L=$PWD/tools/build/libneo_narrow.so
  NEO360_HIP_LIB=$L timeout 300 python tools/bench_train_mlp.py 2>&1 | tail -2 | sed "s/^/$v: /" | tee -a $O/train_ab.log
  NEO360_HIP_LIB=$L timeout 600 python bench.py --workload neo360_train --steps 8 --warmup 2 --cpu-rays 0 > $O/train_$v.json 2> $O/train_$v.err
  python -c "
import json; d=json.load(open('$O/train_$v.json')); print('$v: train step ms %.1f' % d['ms_per_step'], {k: round(x, 1) for k, x in d['phases_ms'].items()}, 'rays/s %.0f' % d['value'])" | tee -a $O/train_ab.log
done
