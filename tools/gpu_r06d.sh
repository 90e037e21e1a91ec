# round 6: the tri-plane tap cache of k_tp_mlp_hp (NEO_TP_PLCACHE) - parity, bitwise A/B against the uncached pipeline, time + joules
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06d; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
B=$PWD/tools/build
L=$O/plcache_ab.log
micro() { NEO360_HIP_LIB=$2 POLL=${3:-1} PREC=f16x3 R=8192 SLOTS=1,0 REPS=400 TAG=$1 timeout 200 python tools/bench_tp_kernel.py 2>&1 | grep -v amdgpu.ids | tee -a $L; }
micro plcache1 ""
micro plcache0 $B/libneo_plc0.so
micro plcache1b ""
micro plcache0b $B/libneo_plc0.so
micro ablate1024 $B/libneo_ablate1024.so 0
micro ablate2048 $B/libneo_ablate2048.so 0
timeout 1200 python -m pytest tests/test_gpu_neo360.py tests/test_gpu_neo360_stages.py tests/test_gpu_repeatable.py tests/test_gpu_edges.py tests/test_gpu_host_r6.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest.log 2>&1; tail -8 $O/pytest.log
frame() { NEO360_HIP_LIB=$2 timeout 300 python bench.py --steps 8 --warmup 2 --cpu-rays 0 --others 0 --exact-f32 0 --setup-timing 0 --chunk-loop 0 > $O/frame_$1.json 2> $O/frame_$1.err
  python - "$1" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.load(open("gpurun_out/r06d/frame_%s.json" % tag)); r = d["roofline"]
    k = {n: round(v["avg_launch_ms"], 2) for n, v in r["kernels"].items()}
    print("%-10s rays/s %8.0f  ms %.1f  %s  sclk %.0f  power %.0f W  energy/step %s J" % (tag, d["value"], d["ms_per_step"], k, r["sclk_mhz_mean"], r["power_w_mean"], r.get("energy_j_per_step")))
except Exception as e:
    print(tag, "failed", e)
PY
}
frame plcache1 "" | tee -a $L
frame plcache0 $B/libneo_plc0.so | tee -a $L
frame plcache1b "" | tee -a $L
