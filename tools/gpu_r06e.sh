# round 6: tap-ring depth of the uncached gather pipeline (3 = shipped; 4 fits in 250 VGPRs since the round-4 fold; 5 spills 17 dwords)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06e; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
B=$PWD/tools/build
L=$O/ring_depth.log
micro() { NEO360_HIP_LIB=$2 POLL=1 PREC=f16x3 R=8192 SLOTS=1,0 REPS=400 TAG=$1 timeout 200 python tools/bench_tp_kernel.py 2>&1 | grep -v amdgpu.ids | tee -a $L; }
micro ring3 $B/libneo_plc0.so
micro ring4 $B/libneo_ring4.so
micro ring5 $B/libneo_ring5.so
micro ring3b $B/libneo_plc0.so
micro ring4b $B/libneo_ring4.so
