#!/bin/bash
# round 2: full default bench line + rocprofv3 stats + PMC passes of the same command
OUT=gpurun_out/r02h; mkdir -p $OUT
timeout 900 python bench.py --steps 5 --warmup 1 > $OUT/bench_neo360_f16x3.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench_neo360_f16x3.json
bash tools/pmc_bench.sh neo360 f16x3 k_tp_mlp_hp 2>&1 | tail -40
