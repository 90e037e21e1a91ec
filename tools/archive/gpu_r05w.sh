# counters of the exact-fp32 NeO-360 evaluator (k_tp_mlp) on the bench workload
cd $GRAFT_REPO_ROOT
bash tools/pmc_bench.sh neo360 f32 "k_tp_mlp<" > gpurun_out/pmc_f32.log 2>&1; tail -45 gpurun_out/pmc_f32.log
