# round 6: counter passes of the bench command with / without the ray-patch tile order (HBM-side bytes per launch)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06h; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for g in 1 0; do
  NEO360_RAY_GRID=$g bash tools/pmc_bench.sh neo360 f16x3 "k_tp_mlp_hp<" "k_tp_mlp_hpp<" > $O/pmc_grid$g.log 2>&1
  cp gpurun_out/pmc_neo360_f16x3/summary.json $O/pmc_grid${g}_hp.json; cp gpurun_out/pmc_neo360_f16x3/summary_2.json $O/pmc_grid${g}_hpp.json
  cp gpurun_out/pmc_neo360_f16x3/kernel_stats.csv $O/kernel_stats_grid$g.csv
  python - $g <<'PY'
import json, sys
g = sys.argv[1]
for k in ("hp", "hpp"):
    d = json.load(open("gpurun_out/r06h/pmc_grid%s_%s.json" % (g, k))); dd = d.get("derived", {})
    print("grid", g, k, "launches", d.get("launches_per_pass"), {x: dd.get(x) for x in ("hbm_bytes_per_launch", "mfma_busy_frac", "l2_hit_rate", "tcp_tcc_read_req_per_launch", "lds_conflict_frac", "valu_per_mfma", "sclk_mhz")})
PY
done
