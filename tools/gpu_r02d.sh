#!/bin/bash
OUT=gpurun_out/r02d; mkdir -p $OUT
for ST in 0 100 200 300 500 800; do
  for SLOT in 1 3; do
    TAG=stagger_$ST NEO_TP_STAGGER=$ST SLOT=$SLOT REPS=5 timeout 300 python tools/bench_tp_kernel.py 2>&1 | tail -1 | tee -a $OUT/stagger.log
  done
done
