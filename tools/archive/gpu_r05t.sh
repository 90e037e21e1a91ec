cd $GRAFT_REPO_ROOT; O=gpurun_out/r05t; rm -rf $O; mkdir -p $O
timeout 600 python tools/diag_mip_train.py > $O/diag.log 2>&1; tail -50 $O/diag.log
