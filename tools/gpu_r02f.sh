#!/bin/bash
OUT=gpurun_out/r02f; mkdir -p $OUT
for NWV in 4 8; do
  for SLOT in 1 3; do
    TAG=waves_$NWV NEO_TP_WAVES=$NWV SLOT=$SLOT REPS=5 timeout 300 python tools/bench_tp_kernel.py 2>&1 | tail -1 | tee -a $OUT/waves.log
  done
done
NEO_TP_WAVES=8 timeout 600 python -m pytest tests/test_gpu_neo360.py tests/test_gpu_neo360_stages.py tests/test_gpu_repeatable.py tests/test_gpu_fullsize.py tests/test_gpu_stages.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -6
