"""Summarise rocprofv3 --pmc passes for one kernel (substring match): per-launch MEAN of every counter over all
dispatches of that kernel found in the *_counter_collection.csv files under a directory, plus derived rates.
usage: pmc_summarize.py <dir> <kernel-substring> [--note TEXT]"""
import csv, glob, json, os, sys
root, pat = sys.argv[1], sys.argv[2]
note = sys.argv[4] if len(sys.argv) > 4 and sys.argv[3] == "--note" else ""
mean, dur, nl = {}, [], 0
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    per = {}
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]:
            continue
        d = per.setdefault(int(r["Dispatch_Id"]), {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if r.get("End_Timestamp") and r.get("Start_Timestamp"):
            d["_t"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if not per:
        continue
    nl = max(nl, len(per))
    names = set(k for d in per.values() for k in d)
    for k in names:
        vals = [d[k] for d in per.values() if k in d]
        if k == "_t":
            dur.append(sum(vals) / len(vals))
        else:
            mean[k] = sum(vals) / len(vals)
c = mean
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    from bench import kernel_source_hash          # the kernel sources the passes ran on: bench.py drops summaries of other sources
    wl = os.environ.get("PMC_WORKLOAD") or {"k_tp_mlp": "neo360", "k_vanilla": "vanilla", "k_mip": "mip360", "k_pix": "pixelnerf"}[
        next(k for k in ("k_tp_mlp", "k_vanilla", "k_mip", "k_pix") if pat.startswith(k))]
    src_hash = kernel_source_hash(wl)
except Exception:
    src_hash = None
out = {"kernel_match": pat, "launches_per_pass": nl, "note": note, "kernel_source_sha16": src_hash, "mean_per_launch": c, "derived": {}}
d = out["derived"]
if dur:
    ms = d["avg_launch_ms_under_pmc"] = sum(dur) / len(dur)
if "GRBM_GUI_ACTIVE" in c and dur:
    d["clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / (ms * 1e6)
if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
    d["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)   # 256 CUs x 4 SIMDs
if c.get("SQ_INSTS_MFMA") and "SQ_INSTS_VALU" in c:
    d["valu_per_mfma"] = (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"]
if c.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in c:
    d["lds_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
if "TCP_TCC_READ_REQ_sum" in c and dur:
    d["l2_to_l1_TBps_at_64B_per_req"] = c["TCP_TCC_READ_REQ_sum"] * 64 / (ms * 1e-3) / 1e12
if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]):
    d["l2_hit_frac"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # KiB as reported; FETCH_SIZE doubled for gfx950 (MI355X_MICROARCH.md, HBM section)
    d["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    if dur:
        d["hbm_TBps"] = d["hbm_bytes_per_launch"] / (ms * 1e-3) / 1e12
print(json.dumps(out, indent=1))
