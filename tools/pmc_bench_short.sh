#!/bin/bash
# usage: pmc_bench_short.sh <workload> <precision> <kernel-substring>: pmc_bench.sh for a short GPU slot - counter passes in order of
# importance (matrix / VALU, LDS / L2, then HBM fetch / write), the summary rewritten after every pass, the stats pass last
W=$1; P=$2; K=$3
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_${W}_${P}
rm -rf $OUT; mkdir -p $OUT
CMD="python $REPO/bench.py --workload $W --precision $P --steps 1 --warmup 0 --cpu-rays 0 --others 0 --exact-f32 0 --setup-timing 0 --chunk-loop 0"
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 110 rocprofv3 --kernel-trace --pmc $C -f csv -d $OUT/pass$i -- $CMD > $OUT/pass$i.log 2>&1
  PMC_WORKLOAD=$W python $REPO/tools/pmc_summarize.py $OUT "$K" --note "rocprofv3 --kernel-trace --pmc <pass> -f csv -- $CMD ($i counter passes, no other trace domains)" > $OUT/summary.json
done
cat $OUT/summary.json
timeout 90 rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -6 $OUT/kernel_stats.csv
find $OUT -name "*.csv" -size +1M -delete
find $OUT -name "*agent_info.csv" -delete
