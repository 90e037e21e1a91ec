#!/bin/bash
OUT=gpurun_out/r02g; mkdir -p $OUT
for V in default d2 d3 w3r2 w3r2d3; do
  for SLOT in 1 3; do
    if [ $V = default ]; then L=""; else L=tools/build/libneo_$V.so; fi
    TAG=$V NEO360_HIP_LIB=$L SLOT=$SLOT REPS=5 timeout 300 python tools/bench_tp_kernel.py 2>&1 | tail -1 | tee -a $OUT/variants.log
  done
done
