# round 4: PixelNeRF evaluator with the positional encoding in pair order (one argument reduction per (sin, shifted sin) pair): parity + A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04p; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pixelnerf.py tests/test_gpu_repeatable.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for rep in 1 2; do for v in pixbase default; do
LIB=""; [ $v = pixbase ] && LIB=tools/build/libneo_pixbase.so
NEO360_HIP_LIB=$LIB timeout 600 python bench.py --workload pixelnerf --steps 10 --warmup 3 --cpu-rays 0 --others 0 --exact-f32 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v pixelnerf rays/s %.0f ms %.2f' % (d['value'], d['ms_per_step']), 'avg launch %.2f ms' % r['avg_launch_ms'], 'alg TFLOP/s %.1f' % r['achieved'], round(r.get('sclk_mhz_mean') or 0), round(r.get('power_w_mean') or 0))" | tee -a $O/pix_ab.log
done; done
