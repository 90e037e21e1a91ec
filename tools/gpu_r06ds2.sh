# round 6: the tree with the direction-sum table in the exact-fp32 evaluator: f32 frame rate, smoke(), the test files that drive the exact kernels
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06ds2; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 60 python bench.py --workload neo360 --precision f32 --steps 3 --warmup 1 --others 0 --exact-f32 0 --chunk-loop 0 --cpu-rays 0 --setup-timing 0 > $O/bench_f32.json 2> $O/bench_f32.err
python -c "
import json; d=json.load(open('$O/bench_f32.json')); r=d['roofline']; print('f32 frame', d['value'], d['ms_per_step'], r.get('frac_executed'), r.get('executed_tflops'))"
timeout 40 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log
timeout 100 python -m pytest tests/test_gpu_repeatable.py tests/test_gpu_range_guard.py tests/test_gpu_host_r4.py tests/test_gpu_flags.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
