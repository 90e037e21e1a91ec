# round 5: weight-gradient kernel k_dw (128 x 128 tiles, bias gradient fused) - gradient tests, the training-op micro-benchmark
# and the training-step line, A/B against k_sgemm<true, true> + k_colsum
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05l; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_host_r4.py tests/test_gpu_host_r5.py -q -m gpu -k "train or grad or backward or differentiable or projected or vanilla" --durations=3 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
for v in dw nodw; do
  L=""; [ $v = nodw ] && L=$PWD/tools/build/libneo_nodw.so
  NEO360_HIP_LIB=$L timeout 300 python tools/bench_train_mlp.py 2>&1 | tail -2 | sed "s/^/$v: /" | tee -a $O/train_ab.log
  NEO360_HIP_LIB=$L timeout 600 python bench.py --workload neo360_train --steps 8 --warmup 2 --cpu-rays 0 > $O/train_$v.json 2> $O/train_$v.err
  python -c "
import json; d=json.load(open('$O/train_$v.json')); print('$v: train step ms %.1f' % d['ms_per_step'], {k: round(x, 1) for k, x in d['phases_ms'].items()}, 'rays/s %.0f' % d['value'])" | tee -a $O/train_ab.log
done
