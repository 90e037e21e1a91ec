# round 6: 2-D ray-patch tile order (neo_ctx_set_ray_grid) - bitwise test, frames with / without the hint (time, joules), then
# the counter passes of the bench command (HBM-side bytes per launch with the hint)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06g; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_host_r6.py tests/test_gpu_fullsize.py::test_neo360_full_frame_properties -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log
L=$O/ray_grid_ab.log
frame() { NEO360_RAY_GRID=$2 timeout 300 python bench.py --steps 8 --warmup 2 --cpu-rays 0 --others 0 --exact-f32 0 --setup-timing 0 --chunk-loop 0 > $O/frame_$1.json 2> $O/frame_$1.err
  python - "$1" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.load(open("gpurun_out/r06g/frame_%s.json" % tag)); r = d["roofline"]
    k = {n: round(v["avg_launch_ms"], 2) for n, v in r["kernels"].items()}
    print("%-10s rays/s %8.0f  ms %.1f  %s  sclk %.0f  power %.0f W  energy/step %.1f J" % (tag, d["value"], d["ms_per_step"], k, r["sclk_mhz_mean"], r["power_w_mean"], r.get("energy_j_per_step") or 0))
except Exception as e:
    print(tag, "failed", e)
PY
}
frame grid1 1 | tee -a $L
frame grid0 0 | tee -a $L
frame grid1b 1 | tee -a $L
frame grid0b 0 | tee -a $L
for g in 1 0; do
  NEO360_RAY_GRID=$g PMC_TAG=grid$g bash tools/pmc_bench.sh neo360 f16x3 "k_tp_mlp_hp<" "k_tp_mlp_hpp<" > $O/pmc_grid$g.log 2>&1
  for f in gpurun_out/pmc_neo360_f16x3*/summary*.json; do [ -f "$f" ] && cp "$f" $O/pmc_grid${g}_$(basename $(dirname $f))_$(basename $f); done
  tail -5 $O/pmc_grid$g.log
done
