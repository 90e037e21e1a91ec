#!/bin/bash
OUT=gpurun_out/r02e; mkdir -p $OUT
for PAD in 0 40000; do
  TAG=ldspad_$PAD NEO_TP_LDS_PAD=$PAD SLOT=1 REPS=5 timeout 300 python tools/bench_tp_kernel.py 2>&1 | tail -1 | tee -a $OUT/occ.log
done
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for V in hp noproj; do
  if [ $V = noproj ]; then export NEO360_TP_PREPROJECT=0; fi
  SLOT=1 REPS=1 timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES -f csv -d $REPO/$OUT/ic_$V -- python $REPO/tools/bench_tp_kernel.py > $REPO/$OUT/ic_$V.log 2>&1
  python $REPO/tools/pmc_summarize.py $REPO/$OUT/ic_$V k_tp_mlp | head -30
done
find $REPO/$OUT -name "*.csv" -size +2M -delete
