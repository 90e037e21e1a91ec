# NOTE: the NEO_TP32_STAGGER switch this script drives was removed after the measurement (no effect; profiles/r06_f32_stagger.log)
# round 6: exact-fp32 NeO-360 evaluator - start offset of the second workgroup generation (anti-phase of the two co-resident workgroups)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06k; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
L=$O/f32_stagger.log
for st in 0 400 800 1200 1600 2400 0; do
  NEO_TP32_STAGGER=$st PREC=f32 R=8192 SLOTS=1,3 REPS=60 TAG=stagger$st timeout 200 python tools/bench_tp_kernel.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee -a $L
done
