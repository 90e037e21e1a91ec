# kernel-level anatomy of the training step (rocprofv3 --kernel-trace --stats of bench.py --workload neo360_train)
O=$GRAFT_REPO_ROOT/gpurun_out/r05e; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/stats -- python $GRAFT_REPO_ROOT/bench.py --workload neo360_train --steps 3 --warmup 1 --cpu-rays 0 > $O/run.log 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
head -32 $O/kernel_stats.csv | cut -c1-220
find $O/stats -name "*.csv" -size +1M -delete
