# NOTE: the NEO_TP32_PAIRS switch and the lib/ab/ variant libraries this script compares were removed after the measurement (the paired encoding is the shipped code; profiles/r06_f32_pairs.log)
# round 6: exact-fp32 evaluator, outside launches (C = 4): pairs of the first stage in the prologue + singles (hybrid) against feature by feature
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06pe2; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
L=$O/f32_pairs_c4.log
AB=$GRAFT_REPO_ROOT/neo-360_amd/lib/ab
for rep in 1 2; do
  PREC=f32 R=8192 SLOTS=3,2,1 REPS=40 TAG=hybrid timeout 150 python tools/bench_tp_kernel.py 2>&1 | grep -v amdgpu.ids | cut -c1-220 | tee -a $L
  NEO360_HIP_LIB=$AB/libneo360_hip_p0.so PREC=f32 R=8192 SLOTS=3,2,1 REPS=40 TAG=single timeout 150 python tools/bench_tp_kernel.py 2>&1 | grep -v amdgpu.ids | cut -c1-220 | tee -a $L
done
timeout 150 python -m pytest tests/test_gpu_neo360_stages.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
