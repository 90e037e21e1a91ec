# round 6: (1) call-overlap tests + the suites that exercise stream ordering, (2) the chunk-loop leg of bench.py,
# (3) gather-half ablations 256 (every tap -> texel 0) / 512 (no loads, blends kept) as isolated test launches
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06b; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
B=$PWD/tools/build
timeout 900 python -m pytest tests/test_gpu_host_r6.py tests/test_gpu_host_r4.py tests/test_gpu_host_r5.py tests/test_gpu_neo360.py tests/test_gpu_multirank.py -x -q -m gpu > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout 600 python bench.py --steps 6 --warmup 2 --cpu-rays 0 --others 0 --exact-f32 0 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06b/bench.json"))
print("headline", d["value"], "chunk_loop", json.dumps(d.get("chunk_loop"), indent=1))
PY
L=$O/gather_ablations.log
micro() { NEO360_HIP_LIB=$2 POLL=0 PREC=f16x3 R=8192 SLOTS=1,0 REPS=500 TAG=$1 timeout 200 python tools/bench_tp_kernel.py 2>/dev/null | tee -a $L; }
micro default ""
micro ablate256 $B/libneo_ablate256.so
micro ablate512 $B/libneo_ablate512.so
micro ablate3 $B/libneo_ablate3.so
micro default2 ""
