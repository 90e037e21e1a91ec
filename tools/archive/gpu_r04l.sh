# round 4: phase trace (s_memtime sums per phase, wave 0 of every workgroup) of both NeO-360 evaluators, library built -DNEO_TP_TRACE=1
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04l; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
export NEO360_HIP_LIB=tools/build/libneo_trace.so
for sl in 3 2; do PP=3 TAG=trace SLOT=$sl N=$([ $sl = 3 ] && echo 385 || echo 129) REPS=10 TRACE=hpp timeout 300 python tools/bench_tp_kernel.py 2>&1 | grep -v amdgpu >> $O/trace.log; done
for sl in 1 0; do PP=3 TAG=trace SLOT=$sl N=$([ $sl = 1 ] && echo 385 || echo 129) REPS=10 TRACE=1 timeout 300 python tools/bench_tp_kernel.py 2>&1 | grep -v amdgpu >> $O/trace.log; done
cat $O/trace.log
