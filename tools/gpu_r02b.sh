#!/bin/bash
OUT=gpurun_out/r02b; mkdir -p $OUT
(cd /tmp && rocprofv3 -L > $OLDPWD/$OUT/counters.txt 2>&1); wc -l $OUT/counters.txt
timeout 300 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_range_guard.py -m gpu -q -s -p no:cacheprovider -k "full_frame_properties or scaled_features" 2>&1 | tail -12
timeout 120 tools/build/pk_f32_repro 400 4096 3 > $OUT/pk_f32_repro.log 2>&1; tail -25 $OUT/pk_f32_repro.log
bash tools/pmc_tp2.sh r02b/pmc_hp_fg SLOT=1 TAG=hp_fg
bash tools/pmc_tp2.sh r02b/pmc_hp_bg SLOT=3 TAG=hp_bg
