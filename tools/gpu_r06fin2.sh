# round 6: last verification of the final tree (paired encoding in the exact-fp32 evaluator): the driver's bench command, smoke(), the whole GPU suite
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06fin2; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json")); r = d["roofline"]
print("rays/s %.0f ms %.1f frac %.3f traffic %s power %s sclk %s" % (d["value"], d["ms_per_step"], r["frac"], r.get("traffic"), r.get("power_w_mean"), r.get("sclk_mhz_mean")))
print("chunk_loop", d["chunk_loop"]["value"], d["chunk_loop"]["frac_of_headline"], "train", d["training_step"]["ms_per_step"], "exact", d["exact_f32"]["value"])
print({k: round(v["value"]) for k, v in d["other_workloads"].items()}, d["scene_setup"]["runs_ms"])
PY

python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 800 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
