# round 6: the whole -m gpu suite on the tree with lanes / overlap / advisor fixes / own texel-space GEMM, then the training step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06c; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
cp gpurun_out/parity_report.json $O/parity_report.json 2>/dev/null
timeout 300 python bench.py --workload neo360_train --steps 10 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err; tail -2 $O/bench_train.err; head -c 1500 $O/bench_train.json; echo
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_train -o train -- python $GRAFT_REPO_ROOT/bench.py --workload neo360_train --steps 5 --warmup 2 --cpu-rays 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $O/prof_train -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" > $O/train_kernel_stats_head.csv && cat $O/train_kernel_stats_head.csv | cut -c1-150
