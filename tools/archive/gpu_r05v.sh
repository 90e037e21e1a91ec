cd $GRAFT_REPO_ROOT; O=gpurun_out/r05v; rm -rf $O; mkdir -p $O
timeout 900 python tools/bench_train_other.py > $O/train_other.log 2>&1; tail -8 $O/train_other.log
RAYS=4096 timeout 900 python tools/bench_train_other.py >> $O/train_other.log 2>&1; tail -4 $O/train_other.log
