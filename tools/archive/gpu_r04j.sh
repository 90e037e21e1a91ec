# round 4: per-kernel times of one Mip-NeRF 360 frame (64/64/128) with the layered NeRF MLP: rocprofv3 kernel stats
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04j; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --workload mip360_128 --steps 1 --warmup 1 --cpu-rays 0 --others 0 --exact-f32 0 > $GRAFT_REPO_ROOT/$O/bench.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); echo $f
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print("%-90s calls %6s  total %9.2f ms  avg %9.1f us  %5.1f %%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
cp $f $O/kernel_stats.csv
