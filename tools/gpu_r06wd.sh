# round 6: exact-fp32 evaluator - weight fragments two chunks ahead of the MFMAs (NEO_TP32_WDEPTH = 2) against one ahead (previous library):
# bitwise check, then the timing A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06wd; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
L=$O/f32_wdepth.log
PREV=$GRAFT_REPO_ROOT/neo-360_amd/lib/ab/libneo360_hip_prev.so
(OUT=$O/new.pt timeout 40 python tools/f32_dirsum_check.py 2>&1 | grep -v amdgpu.ids > $O/c1.log) &
(NEO360_HIP_LIB=$PREV OUT=$O/prev.pt timeout 40 python tools/f32_dirsum_check.py 2>&1 | grep -v amdgpu.ids > $O/c2.log) &
wait
cat $O/c1.log $O/c2.log | tee -a $L
python tools/f32_dirsum_check.py --compare $O/new.pt $O/prev.pt 2>&1 | tee -a $L
rm -f $O/new.pt $O/prev.pt
PREC=f32 R=8192 SLOTS=1,3,0,2 REPS=25 TAG=depth2 timeout 40 python tools/bench_tp_kernel.py 2>&1 | grep -v amdgpu.ids | cut -c1-220 | tee -a $L
NEO360_HIP_LIB=$PREV PREC=f32 R=8192 SLOTS=1,3,0,2 REPS=25 TAG=depth1 timeout 40 python tools/bench_tp_kernel.py 2>&1 | grep -v amdgpu.ids | cut -c1-220 | tee -a $L
