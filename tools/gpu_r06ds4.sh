cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/pmc_neo360_f32_ds; rm -rf $OUT; mkdir -p $OUT
CMD="python $REPO/bench.py --workload neo360 --precision f32 --steps 1 --warmup 0 --cpu-rays 0 --others 0 --exact-f32 0 --setup-timing 0 --chunk-loop 0"
timeout 30 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS -f csv -d $OUT/pass1 -- $CMD > $OUT/pass1.log 2>&1
PMC_WORKLOAD=neo360 python $REPO/tools/pmc_summarize.py $OUT "k_tp_mlp<" --note "one counter pass (matrix / VALU), tree with the direction-sum table" > $OUT/summary.json
cat $OUT/summary.json | grep -E "mfma_busy|valu_per|avg_launch|SQ_INSTS_VALU"
timeout 25 rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -4 $OUT/kernel_stats.csv | cut -c1-60,330-420
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*agent_info.csv" -delete
