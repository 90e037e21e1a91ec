# round 6: patch shape of the ray-patch tile order (HBM-side bytes per launch, FETCH_SIZE pass only) + the new overlap tests
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06i; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_host_r6.py tests/test_gpu_vanilla.py tests/test_gpu_pixelnerf.py tests/test_gpu_mip360.py -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-rays 0 --others 0 --exact-f32 0 --setup-timing 0 --chunk-loop 0"
for shape in 3,3 2,2 1,2 1,1 2,1 1,3 0,3 0,2; do
  d=$GRAFT_REPO_ROOT/$O/fetch_${shape/,/x}; rm -rf $d
  NEO_TP_PATCH=$shape timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $d -- $CMD > $d.log 2>&1
  python - $d $shape <<'PY'
import csv, glob, sys
d, shape = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = {}
for r in rows:
    k = r["Kernel_Name"]
    if "k_tp_mlp_hp" in k and r["Counter_Name"] == "FETCH_SIZE":
        name = "hpp" if "k_tp_mlp_hpp" in k else "hp"
        agg.setdefault(name, []).append(float(r["Counter_Value"]))
print("patch 2^(%s)" % shape, {k: "%.1f GB fetch x2 per launch (%d launches)" % (2 * sum(v) / len(v) * 1024 / 1e9, len(v)) for k, v in agg.items()})
PY
done
