#!/bin/bash
# Instruction-fetch counters of the NeO-360 evaluator micro-benchmark (counters only: --kernel-trace + --pmc).
# usage: pmc_icache.sh <out-subdir> [env assignments for the benchmark...]
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$1; shift
rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_PREFETCH|SQC_" | head -40 > $OUT/available.txt
i=0
while read -r P; do
  [ -z "$P" ] && continue
  i=$((i+1))
  env "$@" REPS=1 timeout 300 rocprofv3 --kernel-trace --pmc $P -f csv -d $OUT/pass$i -- python $REPO/tools/bench_tp_kernel.py > $OUT/pass$i.log 2>&1
  echo "pass$i [$P]: $(tail -1 $OUT/pass$i.log | cut -c1-160)"
done <<'PASSES'
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU
PASSES
python $REPO/tools/pmc_summarize.py $OUT k_tp_mlp > $OUT/summary.json
cat $OUT/summary.json
cat $OUT/available.txt
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*agent_info.csv" -delete
