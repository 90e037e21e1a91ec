# round 6: exact-fp32 evaluator - view-summed direction encodings from the per-ray table (k_tp_dirsum) instead of per point and view:
# bitwise check against the previous library (whole-batch and chunked direction tiling), then the timing A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06ds; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
L=$O/f32_dirsum.log
PREV=$GRAFT_REPO_ROOT/neo-360_amd/lib/ab/libneo360_hip_prev.so
OUT=$O/new.pt timeout 100 python tools/f32_dirsum_check.py 2>&1 | grep -v amdgpu.ids | tee -a $L
NEO360_HIP_LIB=$PREV OUT=$O/prev.pt timeout 100 python tools/f32_dirsum_check.py 2>&1 | grep -v amdgpu.ids | tee -a $L
python tools/f32_dirsum_check.py --compare $O/new.pt $O/prev.pt 2>&1 | tee -a $L
rm -f $O/new.pt $O/prev.pt
for rep in 1 2; do
  PREC=f32 R=8192 SLOTS=1,3,0,2 REPS=30 TAG=table timeout 100 python tools/bench_tp_kernel.py 2>&1 | grep -v amdgpu.ids | cut -c1-220 | tee -a $L
  NEO360_HIP_LIB=$PREV PREC=f32 R=8192 SLOTS=1,3,0,2 REPS=30 TAG=perview timeout 100 python tools/bench_tp_kernel.py 2>&1 | grep -v amdgpu.ids | cut -c1-220 | tee -a $L
done
