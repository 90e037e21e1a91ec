cd $GRAFT_REPO_ROOT; O=gpurun_out/r06v; rm -rf $O; mkdir -p $O
timeout 300 python tools/prof_train_copies.py 2>&1 | grep -v amdgpu.ids | tee $O/copies.log | tail -45
