# round 4: s_setprio around the L1..L3 k-steps of both NeO-360 evaluators (NEO_TP_PRIO): the four test launches, interleaved
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04k; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do for v in base prio1 prio3; do
  NEO360_HIP_LIB=tools/build/libneo_$v.so PP=3 TAG=$v SLOTS=1,3,0,2 REPS=40 timeout 300 python tools/bench_tp_kernel.py 2>&1 | grep -E "slot" >> $O/kernel_ab.log
done; done
cat $O/kernel_ab.log
for v in base prio1 prio3 base prio1 prio3; do
NEO360_HIP_LIB=tools/build/libneo_$v.so timeout 300 python bench.py --steps 6 --warmup 2 --cpu-rays 0 --others 0 --exact-f32 0 --setup-timing 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v rays/s %.0f ms %.1f' % (d['value'], d['ms_per_step']), {k: round(x['avg_launch_ms'],2) for k,x in r['kernels'].items()}, round(r.get('sclk_mhz_mean') or 0), round(r.get('power_w_mean') or 0))" | tee -a $O/frame_ab.log
done
