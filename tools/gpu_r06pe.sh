# NOTE: the NEO_TP32_PAIRS switch and the lib/ab/ variant libraries this script compares were removed after the measurement (the paired encoding is the shipped code; profiles/r06_f32_pairs.log)
# round 6: exact-fp32 NeO-360 evaluator k_tp_mlp<., 2> - pos_enc by (sin, cos) pairs in the prologue (NEO_TP32_PAIRS) against
# feature by feature; variant libraries are built on the CPU side into neo-360_amd/lib/ab/ (see profiles/r06_f32_pairs.log)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06pe; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
L=$O/f32_pairs.log
AB=$GRAFT_REPO_ROOT/neo-360_amd/lib/ab
for rep in 1 2; do
  PREC=f32 R=8192 SLOTS=1,3,0,2 REPS=40 TAG=pairs timeout 200 python tools/bench_tp_kernel.py 2>&1 | grep -v amdgpu.ids | cut -c1-220 | tee -a $L
  NEO360_HIP_LIB=$AB/libneo360_hip_p0.so PREC=f32 R=8192 SLOTS=1,3,0,2 REPS=40 TAG=single timeout 200 python tools/bench_tp_kernel.py 2>&1 | grep -v amdgpu.ids | cut -c1-220 | tee -a $L
done
NEO360_HIP_LIB=$AB/libneo360_hip_tr.so PREC=f32 R=8192 SLOTS=1,3 REPS=3 TRACE=f32 TAG=pairs timeout 200 python tools/bench_tp_kernel.py 2>&1 | grep -v amdgpu.ids | cut -c1-220 | tee -a $L
NEO360_HIP_LIB=$AB/libneo360_hip_tr0.so PREC=f32 R=8192 SLOTS=1,3 REPS=3 TRACE=f32 TAG=single timeout 200 python tools/bench_tp_kernel.py 2>&1 | grep -v amdgpu.ids | cut -c1-220 | tee -a $L
timeout 600 python -m pytest tests/test_gpu_neo360_stages.py tests/test_gpu_repeatable.py tests/test_gpu_range_guard.py tests/test_gpu_fullsize.py tests/test_gpu_host_r4.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
