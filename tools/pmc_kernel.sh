#!/bin/bash
# Counter passes (--kernel-trace + --pmc only, no other trace domains) over one micro-benchmark script.
# usage: pmc_kernel.sh <out-subdir> <tools/script.py> <kernel-substring> [env assignments...]
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$1; SCRIPT=$2; PAT=$3; shift 3
rm -rf $OUT; mkdir -p $OUT
i=0
while read -r P; do
  [ -z "$P" ] && continue
  i=$((i+1))
  env "$@" REPS=1 timeout 300 rocprofv3 --kernel-trace --pmc $P -f csv -d $OUT/pass$i -- python $REPO/$SCRIPT > $OUT/pass$i.log 2>&1
  echo "pass$i [$P]: $(tail -1 $OUT/pass$i.log | cut -c1-160)"
done <<'PASSES'
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
FETCH_SIZE
WRITE_SIZE
PASSES
python $REPO/tools/pmc_summarize.py $OUT "$PAT" --note "rocprofv3 --kernel-trace --pmc <pass> -f csv -- python $SCRIPT ($*), 4 counter-only passes" > $OUT/summary.json
cat $OUT/summary.json
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*agent_info.csv" -delete
