#!/bin/bash
OUT=gpurun_out/r02c; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_repeatable.py -m gpu -q -p no:cacheprovider -k "full_frame_properties or neo360_evaluators" 2>&1 | tail -4
for SLOT in 1; do
  TAG=base SLOT=$SLOT REPS=5 timeout 300 python tools/bench_tp_kernel.py 2>&1 | tail -1 | tee -a $OUT/ablate.log
  for A in 1 2 3 4 8 16 32 64 128 255; do
    TAG=ablate_$A NEO360_HIP_LIB=tools/build/libneo_abl$A.so SLOT=$SLOT REPS=5 timeout 300 python tools/bench_tp_kernel.py 2>&1 | tail -1 | tee -a $OUT/ablate.log
  done
done
