# round 4: proposal MLPs on 64-row tiles (k_mip_prop_h) against the 32-row evaluator (library variant -DNEO_MIP_PROP64=0), one box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04n; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_mip360.py tests/test_gpu_repeatable.py -q -m gpu -x -k "mip or Mip" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for rep in 1 2; do for v in prop32 default; do
LIB=""; [ $v = prop32 ] && LIB=tools/build/libneo_prop32.so
for wl in mip360 mip360_128; do
NEO360_HIP_LIB=$LIB timeout 600 python bench.py --workload $wl --steps 3 --warmup 1 --cpu-rays 0 --others 0 --exact-f32 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v $wl rays/s %.0f ms %.1f' % (d['value'], d['ms_per_step']), {k[:24]: (round(x['avg_launch_ms'],2), round(x['algorithmic_tflops'],1)) for k,x in r['kernels'].items()}, round(r.get('sclk_mhz_mean') or 0), round(r.get('power_w_mean') or 0))" | tee -a $O/mip_ab.log
done; done; done
