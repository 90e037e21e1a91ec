# round 5: split-K of the weight-gradient GEMMs sized for >= 1024 workgroups (train_mlp.hip:split_k) - gradient tests + training-step A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05i; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_host_r5.py -q -m gpu -k "mlp or training or vanilla" --durations=3 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for v in new old new2; do
  L=""; [ $v = old ] && L=$PWD/tools/build/libneo_train_oldsplit.so
  NEO360_HIP_LIB=$L timeout 600 python bench.py --workload neo360_train --steps 8 --warmup 2 --cpu-rays 0 > $O/train_$v.json 2> $O/train_$v.err
  python -c "
import json; d=json.load(open('$O/train_$v.json')); print('$v: train step ms %.1f' % d['ms_per_step'], {k: round(x, 1) for k, x in d['phases_ms'].items()}, 'rays/s %.0f' % d['value'])" | tee -a $O/train_ab.log
done
python tools/bench_train_mlp.py 2>&1 | tail -8 | tee $O/train_mlp_bench.log
