"""HBM traffic of one NeO-360 training step: sums FETCH_SIZE / WRITE_SIZE (two counter-only rocprofv3 passes over
`bench.py --workload neo360_train --steps S --warmup 0 --cpu-rays 0`) over every kernel of the run and divides by the steps the run
executed (warm-up + timed + the phase-timing step), per kernel family as well.  usage: pmc_train_step.py <dir with pass1 / pass2> <steps>"""
import csv, glob, json, os, sys
root, steps = sys.argv[1], float(sys.argv[2])
tot, fam = {}, {}
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        c, v = r["Counter_Name"], float(r["Counter_Value"])
        if c not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        tot[c] = tot.get(c, 0.0) + v
        n = r["Kernel_Name"]
        key = next((k for k in ("k_tp_chain_fwd", "k_tp_chain_bwd", "k_dw_reduce", "k_dw", "k_sgemm", "k_gather_bwd_runs", "k_map_gather_bwd_runs",
                                "k_map_gather", "k_gather", "k_transpose", "k_view_mean", "at::native") if k in n), "other")
        d = fam.setdefault(key, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
        d[c] += v
gb = lambda d: (2.0 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024.0 / steps / 1e9     # KiB; FETCH_SIZE doubled on gfx950
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    from bench import kernel_source_hash
    sha = kernel_source_hash("neo360_train")
except Exception:
    sha = None
out = {"kernel_source_sha16": sha, "steps_in_run": steps, "hbm_GB_per_step": gb(tot), "read_GB_per_step": 2.0 * tot.get("FETCH_SIZE", 0.0) * 1024.0 / steps / 1e9,
       "write_GB_per_step": tot.get("WRITE_SIZE", 0.0) * 1024.0 / steps / 1e9,
       "by_kernel_GB_per_step": {k: round(gb(v), 3) for k, v in sorted(fam.items(), key=lambda kv: -gb(kv[1]))},
       "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) over bench.py --workload neo360_train; all kernels of the "
               "run (set-up kernels included: a few per cent) / steps executed; KiB units, FETCH_SIZE x 2 (MI355X_MICROARCH.md, HBM section)"}
print(json.dumps(out, indent=1))
