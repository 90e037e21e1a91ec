# phase trace of the exact-fp32 NeO-360 evaluator (k_tp_mlp on the projected maps): s_memtime sums per phase, fine inside / outside launches
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05ad; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for slot in 1 3; do
  NEO360_HIP_LIB=$PWD/tools/build/libneo_tp32trace.so PREC=f32 TRACE=f32 SLOT=$slot R=8192 REPS=2 TAG=slot$slot timeout 300 python tools/bench_tp_kernel.py 2>&1 | grep -v amdgpu | tee -a $O/trace.log
done
