# round 5: k_dw with lane = n in the epilogue (coalesced atomics); ablation: no accumulation into W
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05n; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_host_r5.py -q -m gpu -k "backward or vanilla_training" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for v in dw dwnoat; do
  L=""; [ $v != dw ] && L=$PWD/tools/build/libneo_$v.so
  NEO360_HIP_LIB=$L timeout 300 python tools/bench_train_mlp.py 2>&1 | tail -2 | sed "s/^/$v: /" | tee -a $O/train_ab.log
done
NEO360_HIP_LIB="" timeout 600 python bench.py --workload neo360_train --steps 8 --warmup 2 --cpu-rays 0 > $O/train_dw.json 2> $O/train_dw.err
python -c "
import json; d=json.load(open('$O/train_dw.json')); print('dw: train step ms %.1f' % d['ms_per_step'], {k: round(x, 1) for k, x in d['phases_ms'].items()}, 'rays/s %.0f' % d['value'])" | tee -a $O/train_ab.log
