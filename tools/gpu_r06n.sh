# round 6: tap cache v2 (template parameter PLC of k_tp_mlp_hp; fine launches only) - A/B by $NEO_TP_PLC, then parity + frames
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06n; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
L=$O/plcache_v2.log
micro() { env $2 POLL=1 PREC=f16x3 R=8192 SLOTS=1,0 REPS=300 TAG=$1 timeout 200 python tools/bench_tp_kernel.py 2>&1 | grep -v amdgpu.ids | cut -c1-230 | tee -a $L; }
micro plc_auto "A=1"
micro plc_off "NEO_TP_PLC=0"
micro plc_on "NEO_TP_PLC=1"
micro plc_off2 "NEO_TP_PLC=0"
micro plc_auto2 "A=1"
timeout 900 python -m pytest tests/test_gpu_neo360.py tests/test_gpu_neo360_stages.py tests/test_gpu_repeatable.py tests/test_gpu_host_r6.py -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log
frame() { env $2 timeout 300 python bench.py --steps 8 --warmup 2 --cpu-rays 0 --others 0 --exact-f32 0 --setup-timing 0 --chunk-loop 0 > $O/frame_$1.json 2> $O/frame_$1.err
  python - "$1" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.load(open("gpurun_out/r06n/frame_%s.json" % tag)); r = d["roofline"]
    k = {n: round(v["avg_launch_ms"], 2) for n, v in r["kernels"].items()}
    print("%-10s rays/s %8.0f  ms %.1f  %s  sclk %.0f  power %.0f W  energy/step %.1f J" % (tag, d["value"], d["ms_per_step"], k, r["sclk_mhz_mean"], r["power_w_mean"], r.get("energy_j_per_step") or 0))
except Exception as e:
    print(tag, "failed", e)
PY
}
frame auto "A=1" | tee -a $L
frame off "NEO_TP_PLC=0" | tee -a $L
frame auto2 "A=1" | tee -a $L
frame off2 "NEO_TP_PLC=0" | tee -a $L
