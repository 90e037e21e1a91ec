# round 5, fourth GPU call: the training call on its fused point / activation operators (neo_tp_train_points, neo_tp_activate),
# the distribution rule on the sharp full-size fixture, the training-step line again
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05d; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_host_r5.py tests/test_gpu_training.py tests/test_gpu_fullsize.py tests/test_gpu_vanilla.py -q -m gpu --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 600 python bench.py --workload neo360_train --steps 8 --warmup 2 --cpu-rays 0 > $O/train.json 2> $O/train.err; cut -c1-400 $O/train.json; python -c "
import json; d=json.load(open('$O/train.json')); print('train step ms %.1f' % d['ms_per_step'], d['phases_ms'], 'rays/s %.0f' % d['value'])"; tail -2 $O/train.err
