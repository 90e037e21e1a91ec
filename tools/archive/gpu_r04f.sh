# round 4: the bottleneck fold in the vanilla evaluator (C2) + the full GPU suite on the tree with both folds
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04f; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
for lib in $PWD/tools/build/libneo_vh_nofold.so "" $PWD/tools/build/libneo_vh_nofold.so ""; do NEO360_HIP_LIB=$lib timeout 600 python bench.py --workload vanilla --steps 10 --warmup 3 --cpu-rays 0 --others 0 --exact-f32 0 2> /dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib'[-20:] or 'default(fold)', 'vanilla rays/s %.0f ms %.2f' % (d['value'], d['ms_per_step']), 'kernel avg ms %.2f' % r['avg_launch_ms'], 'alg TFLOP/s %.1f' % r['achieved'], round(r.get('sclk_mhz_mean') or 0), round(r.get('power_w_mean') or 0))"; done
