"""CPU: the bookkeeping of bench.py that does not need a GPU: the source lists the PMC stamps are hashed over exist and are
really what the kernels include; a counter summary is attached to a bench line only when its stamp matches the tree."""
import json
import os
import re

import bench

CSRC = os.path.join(bench.ROOT, "neo-360_amd", "csrc")


def _local_includes(path, seen):
    for name in re.findall(r'#include\s+"([^"]+)"', open(path).read()):
        p = os.path.join(CSRC, name)
        if os.path.exists(p) and name not in seen:
            seen.add(name)
            _local_includes(p, seen)
    return seen


def test_kernel_source_lists_cover_the_kernels_own_includes():
    for workload, names in bench.KERNEL_SOURCES.items():
        for n in names:
            assert os.path.exists(os.path.join(CSRC, n)), (workload, n)
        tu = names[0]
        included = _local_includes(os.path.join(CSRC, tu), set())
        # every project header the translation unit pulls in (transitively) is part of the stamp
        assert included <= set(names), (workload, sorted(included - set(names)))
        assert len(bench.kernel_source_hash(workload)) == 16


def test_pmc_summary_is_dropped_when_its_stamp_is_stale(tmp_path, monkeypatch):
    prof = bench.pmc_profile("neo360", "f16x3")
    import glob
    src = sorted(glob.glob(os.path.join(bench.ROOT, "profiles", "r[0-9][0-9]_pmc_neo360_f16x3.json")))[-1]     # newest round wins
    assert os.path.relpath(src, bench.ROOT) == prof["source"]
    committed = json.load(open(src))
    if committed.get("kernel_source_sha16") == bench.kernel_source_hash("neo360"):
        assert prof.get("mfma_busy_frac") and not prof.get("stale")          # the committed summary belongs to this tree
    else:
        assert prof.get("stale") and "mfma_busy_frac" not in prof
    # a different tree: the same summary must not be attached
    monkeypatch.setattr(bench, "kernel_source_hash", lambda workload="neo360": "0" * 16)
    stale = bench.pmc_profile("neo360", "f16x3")
    assert stale.get("stale") and "mfma_busy_frac" not in stale and "hbm_bytes_per_launch" not in stale
