# round 5, third GPU call: the Mip-NeRF 360 trunk as ONE launch (k_mip_chain_h: slab-local barriers instead of seven kernel
# boundaries) - bitwise check + rates of the stand-alone chain on real and zero operands, the library A/B on both Mip workloads;
# the training-step line (bench.py --workload neo360_train); the b5 / b6 full-size fixtures under the cdf-reach rule.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 300 ./tools/build/gemm_h_bench 16384 40 1 > $O/gemm_chain_real.log 2>&1; cat $O/gemm_chain_real.log
timeout 300 ./tools/build/gemm_h_bench 16384 40 0 > $O/gemm_chain_zero.log 2>&1; tail -7 $O/gemm_chain_zero.log
timeout 900 python -m pytest tests/test_gpu_mip360.py tests/test_gpu_fullsize.py tests/test_gpu_host_r5.py tests/test_gpu_repeatable.py -q -m gpu --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
B=$PWD/tools/build
mip() { # tag lib workload
  NEO360_HIP_LIB=$2 timeout 300 python bench.py --workload $3 --steps 4 --warmup 1 --cpu-rays 0 --others 0 --exact-f32 0 > $O/$3_$1.json 2> $O/$3_$1.err
  python - "$1" "$3" <<'PY'
import json, sys
tag, wl = sys.argv[1], sys.argv[2]
try:
    d = json.load(open("gpurun_out/r05c/%s_%s.json" % (wl, tag))); r = d["roofline"]
    print("%-12s %-10s rays/s %8.0f  ms %.1f  sclk %.0f  power %.0f W  %s" % (wl, tag, d["value"], d["ms_per_step"], r["sclk_mhz_mean"] or 0, r["power_w_mean"] or 0,
          {n[:40]: (round(v["avg_launch_ms"], 2), round(v["algorithmic_tflops"], 1)) for n, v in r["kernels"].items()}))
except Exception as e:
    print(wl, tag, "failed", e)
PY
}
for wl in mip360_128 mip360; do
  mip chain "" $wl | tee -a $O/mip_ab.log
  mip eight $B/libneo_mip_nochain.so $wl | tee -a $O/mip_ab.log
  mip chain2 "" $wl | tee -a $O/mip_ab.log
done
timeout 600 python bench.py --workload neo360_train --steps 5 --warmup 1 > $O/train.json 2> $O/train.err; cut -c1-1500 $O/train.json; tail -3 $O/train.err
