# round 6: tap cache diagnostics - v1 with per-segment prefetch; the LDS staging path alone (no dedup: every tap its own slot)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06m; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
B=$PWD/tools/build
L=$O/plcache_diag.log
micro() { NEO360_HIP_LIB=$2 POLL=1 PREC=f16x3 R=8192 SLOTS=1,0 REPS=300 TAG=$1 timeout 200 python tools/bench_tp_kernel.py 2>&1 | grep -v amdgpu.ids | cut -c1-230 | tee -a $L; }
micro shipped ""
micro plcv1b $B/libneo_plcv1b.so
micro nodedup $B/libneo_plcnodedup.so
micro shipped2 ""
