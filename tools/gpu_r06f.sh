# round 6: the pruned tree (experiment branches resolved) - kernel checksums against the logged ones (bit-identical outputs),
# the whole -m gpu suite, a frame with the chunk-loop leg
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06f; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
PREC=f16x3 R=8192 SLOTS=1,0,3,2 REPS=100 TAG=pruned timeout 200 python tools/bench_tp_kernel.py 2>&1 | grep -v amdgpu.ids | tee $O/checksums.log
echo "expected (profiles/r06_energy_budget_raw.log, r05 logs): slot 1 checksum 5613074.400444, slot 0 1918717.721481"
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
cp gpurun_out/parity_report.json $O/parity_report.json 2>/dev/null
timeout 600 python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06f/bench.json"))
print("headline", d["value"], "ms", d["ms_per_step"], "chunk_loop", d["chunk_loop"]["value"], d["chunk_loop"]["frac_of_headline"], "bitwise", d["chunk_loop"]["bitwise_equal_to_whole_frame_call"])
print("exact_f32", d.get("exact_f32", {}).get("value"), "others", {k: v["value"] for k, v in d.get("other_workloads", {}).items()})
print("cpu", d.get("cpu_baseline"), d.get("parity_vs_cpu"))
PY
