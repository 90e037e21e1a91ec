# round 4: exact-fp32 evaluator with the folded tail, PixelNeRF fold: parity + timing
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04g; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
for pp in 1 3; do NEO360_TP_PREPROJECT=$pp timeout 600 python bench.py --precision f32 --steps 2 --warmup 1 --cpu-rays 0 --others 0 --exact-f32 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('f32 pp$pp rays/s %.0f ms %.1f' % (d['value'], d['ms_per_step']), 'avg launch %.1f' % r['avg_launch_ms'], 'executed TFLOP/s %.1f frac %.3f' % (r.get('executed_tflops') or 0, r.get('frac_executed') or 0), round(r.get('sclk_mhz_mean') or 0), round(r.get('power_w_mean') or 0))"; done
timeout 600 python bench.py --workload pixelnerf --steps 10 --warmup 3 --cpu-rays 0 --others 0 --exact-f32 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('pixelnerf rays/s %.0f ms %.2f' % (d['value'], d['ms_per_step']), 'alg TFLOP/s %.1f' % r['achieved'])"
