# round 5: neo_linear_weight_grad for the texel-space projection's weight gradient
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05p; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_host_r5.py -q -m gpu -k "weight_grad or project or end_to_end or differentiable" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
for v in new new2; do
  timeout 600 python bench.py --workload neo360_train --steps 8 --warmup 2 --cpu-rays 0 > $O/train_$v.json 2> $O/train_$v.err
  python -c "
import json; d=json.load(open('$O/train_$v.json')); print('$v: train step ms %.1f' % d['ms_per_step'], {k: round(x, 1) for k, x in d['phases_ms'].items()}, 'rays/s %.0f' % d['value'])" | tee -a $O/train_ab.log
done
