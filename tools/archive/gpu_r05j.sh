# round 5: projected-space training (texel-space projection of the latent, 256-channel lookup, MLP on `pre`) - tests + step A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05j; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_host_r5.py tests/test_gpu_training.py tests/test_gpu_host_r4.py -q -m gpu -k "projected or training or differentiable or gather or chunked or density" --durations=4 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
for v in proj rows proj2; do
  P=1; [ $v = rows ] && P=0
  NEO360_TRAIN_PROJECTED=$P timeout 600 python bench.py --workload neo360_train --steps 8 --warmup 2 --cpu-rays 0 > $O/train_$v.json 2> $O/train_$v.err
  python -c "
import json; d=json.load(open('$O/train_$v.json')); print('$v: train step ms %.1f' % d['ms_per_step'], {k: round(x, 1) for k, x in d['phases_ms'].items()}, 'rays/s %.0f' % d['value'])" | tee -a $O/train_ab.log
  tail -2 $O/train_$v.err
done
