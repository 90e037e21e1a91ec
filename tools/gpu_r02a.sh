#!/bin/bash
# round 2, GPU call A: correctness of the pre-projected NeO-360 evaluator + variant A/B + first bench line
OUT=gpurun_out/r02a; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q --durations=12 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -40 $OUT/pytest.log
for SLOT in 1 3; do
  for V in default w2r2 w2r4 w3r2 w3r1x; do
    if [ $V = default ]; then L=""; else L=tools/build/libneo_$V.so; fi
    TAG=hp_$V NEO360_HIP_LIB=$L SLOT=$SLOT REPS=5 timeout 300 python tools/bench_tp_kernel.py 2>&1 | tail -1 | tee -a $OUT/kern.log
  done
  TAG=old_noproj NEO360_TP_PREPROJECT=0 SLOT=$SLOT REPS=5 timeout 300 python tools/bench_tp_kernel.py 2>&1 | tail -1 | tee -a $OUT/kern.log
done
timeout 900 python bench.py --steps 3 --warmup 1 > $OUT/bench_neo360.json 2> $OUT/bench_neo360.err; echo "bench rc=$?"
cat $OUT/bench_neo360.json; tail -5 $OUT/bench_neo360.err
