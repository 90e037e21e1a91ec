# round 4: work list written by rank (scatter) instead of searched per lane: outside-sphere test launches + frame, A/B on one box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04m; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do for v in base rank; do
  NEO360_HIP_LIB=tools/build/libneo_$v.so PP=3 TAG=$v SLOTS=3,2 REPS=40 timeout 300 python tools/bench_tp_kernel.py 2>&1 | grep -E "slot" >> $O/kernel_ab.log
done; done
cat $O/kernel_ab.log
NEO360_HIP_LIB=tools/build/libneo_rank.so timeout 900 python -m pytest tests/test_gpu_neo360.py tests/test_gpu_neo360_stages.py -q -m gpu -x 2>&1 | tail -3
for v in base rank base rank; do
NEO360_HIP_LIB=tools/build/libneo_$v.so timeout 300 python bench.py --steps 6 --warmup 2 --cpu-rays 0 --others 0 --exact-f32 0 --setup-timing 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v rays/s %.0f ms %.1f' % (d['value'], d['ms_per_step']), {k: round(x['avg_launch_ms'],2) for k,x in r['kernels'].items()}, round(r.get('sclk_mhz_mean') or 0), round(r.get('power_w_mean') or 0))" | tee -a $O/frame_ab.log
done
