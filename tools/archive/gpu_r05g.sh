# round 5: DYNAMIC instruction counts of the evaluator variants (SQ_INSTS_VALU / SQ_INSTS_MFMA per launch, one counter pass each,
# inside-fine test launch slot 1 and outside-fine slot 3): the round-4 kernels, + doubling-order encodings, all four reductions
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; O=$REPO/gpurun_out/r05g; rm -rf $O; mkdir -p $O
for v in r4 pair2 all; do
  L=""; [ $v != r4 ] && L=$REPO/tools/build/libneo_$v.so
  NEO360_HIP_LIB=$L SLOTS=1,3 REPS=1 PP=3 timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU -f csv -d $O/$v -- python $REPO/tools/bench_tp_kernel.py > $O/$v.log 2>&1
  python - $O/$v $v <<'PY'
import csv, glob, sys, os
root, tag = sys.argv[1], sys.argv[2]
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    per = {}
    for r in csv.DictReader(open(f)):
        if "k_tp_mlp_hp" not in r["Kernel_Name"]: continue
        kn = "hpp" if "k_tp_mlp_hpp" in r["Kernel_Name"] else "hp"
        d = per.setdefault((int(r["Dispatch_Id"]), kn), {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for (did, kn), d in sorted(per.items())[-2:]:
        pts = 8192 * 385
        print("%-6s %-4s dispatch %d: VALU %.0f per point (%.3e), MFMA %.1f per point, VALU/MFMA %.2f, LDS %.1f, VMEM_RD %.1f, SALU %.1f per point (wave-level instructions x 64 lanes / points: per-point counts are per wave instruction / (points / 64))" % (
            tag, kn, did, d["SQ_INSTS_VALU"] / (pts / 64), d["SQ_INSTS_VALU"], d["SQ_INSTS_MFMA"] / (pts / 64), d["SQ_INSTS_VALU"] / d["SQ_INSTS_MFMA"],
            d["SQ_INSTS_LDS"] / (pts / 64), d["SQ_INSTS_VMEM_RD"] / (pts / 64), d["SQ_INSTS_SALU"] / (pts / 64)))
PY
done | tee $O/valu_per_point.log
find $O -name "*.csv" -size +1M -delete
