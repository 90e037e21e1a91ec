# round 4: bottleneck folded into view layer 0 (both evaluators) and the empty-tile-view skip of the outside-sphere evaluator
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04e; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_neo360_stages.py tests/test_gpu_neo360.py tests/test_gpu_repeatable.py tests/test_gpu_fullsize.py tests/test_gpu_range_guard.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
K=$O/kernel_ab.log
run() { NEO360_HIP_LIB=$2 PP=$3 TAG=$1 SLOTS=$4 REPS=40 timeout 300 python tools/bench_tp_kernel.py 2>&1 | grep -E "slot|Error|error" >> $K; }
B=$PWD/tools/build
run nofold  $B/libneo_nofold.so 3 1,3,0,2
run fold    "" 3 1,3,0,2
run se0     $B/libneo_se0.so 2 3,2
run se1     $B/libneo_se1.so 2 3,2
run se3     "" 2 3,2
run nofold  $B/libneo_nofold.so 3 1,3,0,2
run fold    "" 3 1,3,0,2
cat $K
for lib in $B/libneo_nofold.so ""; do NEO360_HIP_LIB=$lib timeout 600 python bench.py --steps 8 --warmup 2 --cpu-rays 0 --others 0 --exact-f32 0 2> /dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib'[-16:] or 'default', 'rays/s %.0f ms %.1f' % (d['value'], d['ms_per_step']), {k: round(v['avg_launch_ms'], 2) for k, v in r['kernels'].items()}, round(r.get('sclk_mhz_mean') or 0), round(r.get('power_w_mean') or 0))"; done
