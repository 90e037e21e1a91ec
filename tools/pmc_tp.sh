#!/bin/bash
# PMC passes over the NeO-360 point-evaluator micro-benchmark (counters only: no tracing domains besides kernel-trace).
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_tp_${PREC:-f16x3}
rm -rf $OUT; mkdir -p $OUT
P1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"
P4="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  REPS=1 timeout 600 rocprofv3 --kernel-trace --pmc $P -f csv -d $OUT/pass$i -- python $REPO/tools/bench_tp_kernel.py > $OUT/pass$i.log 2>&1
  tail -1 $OUT/pass$i.log
done
python $REPO/tools/pmc_summarize.py $OUT k_tp_mlp > $OUT/summary.json
cat $OUT/summary.json
find $OUT -name "*.csv" -size +2M -delete
