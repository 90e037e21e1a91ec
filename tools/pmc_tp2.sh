#!/bin/bash
# PMC passes over the NeO-360 point-evaluator micro-benchmark (counters only: --kernel-trace + --pmc, no other domains).
# usage: pmc_tp2.sh <out-subdir> [env assignments for the benchmark...]
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$1; shift
rm -rf $OUT; mkdir -p $OUT
i=0
while read -r P; do
  [ -z "$P" ] && continue
  i=$((i+1))
  env "$@" REPS=1 timeout 300 rocprofv3 --kernel-trace --pmc $P -f csv -d $OUT/pass$i -- python $REPO/tools/bench_tp_kernel.py > $OUT/pass$i.log 2>&1
  echo "pass$i [$P]: $(tail -1 $OUT/pass$i.log | cut -c1-160)"
done <<'PASSES'
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
FETCH_SIZE
WRITE_SIZE
TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum
SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES
PASSES
python $REPO/tools/pmc_summarize.py $OUT k_tp_mlp > $OUT/summary.json
cat $OUT/summary.json
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*agent_info.csv" -delete
