# round 4, task 8: the layer-by-layer NeRF MLP inside the library: parity (mip tests) and the two Mip-NeRF 360 bench lines
# with the fused evaluator (NEO360_MIP_LAYERED=0) and the layered schedule (default) on one box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04i; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mip360.py -q -m gpu -x --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
for wl in mip360_128 mip360; do for lay in 0 1 0 1; do
NEO360_MIP_LAYERED=$lay timeout 600 python bench.py --workload $wl --steps 3 --warmup 1 --cpu-rays 0 --others 0 --exact-f32 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$wl layered=$lay rays/s %.0f ms %.1f' % (d['value'], d['ms_per_step']), '|', r.get('kernel'), 'avg launch %.2f ms' % r['avg_launch_ms'], 'alg TFLOP/s %.1f' % r['achieved'], 'frac split %.3f' % (r.get('frac_of_split_ceiling') or 0), round(r.get('sclk_mhz_mean') or 0), 'MHz', round(r.get('power_w_mean') or 0), 'W')
for k, v in r['kernels'].items(): print('    ', k, v['launches'], '%.3f ms' % v['avg_launch_ms'], '%.1f TF' % v['algorithmic_tflops'])" | tee -a $O/mip_bench.log
done; done
