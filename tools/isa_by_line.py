"""Per source line instruction counts of one kernel (build the .s with -gline-tables-only):

    python tools/isa_by_line.py tp_hp_g.s k_tp_mlp_hpILi3E [min_count]

Buckets every instruction under the last .loc directive (the innermost inlined callee's file:line) and prints VALU / MFMA /
LDS / VMEM / SALU counts per file:line, largest VALU first.  Static counts: code inside the view loop runs once per view."""
import collections
import re
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from isa_segments import classify


def main():
    path, key = sys.argv[1], sys.argv[2]
    floor = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    lines = open(path).read().splitlines()
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]*)"', l)
        if m:
            files[int(m.group(1))] = m.group(2).rsplit("/", 1)[-1]
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and "@" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    cur = ("?", 0)
    tab = collections.defaultdict(collections.Counter)
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        if not t or t[0] in ".;/" or t.endswith(":"):
            continue
        tab[cur][classify(t.split()[0])] += 1
    tot = collections.Counter()
    print("%-22s %5s %5s %5s %5s %5s %5s" % ("file:line", "valu", "mfma", "lds", "vmem", "salu", "wait"))
    for (f, ln), c in sorted(tab.items(), key=lambda kv: -kv[1]["valu"]):
        tot.update(c)
        if c["valu"] >= floor:
            print("%-22s %5d %5d %5d %5d %5d %5d" % ("%s:%d" % (f, ln), c["valu"], c["mfma"], c["lds"], c["vmem"], c["salu"], c["wait"]))
    print("total", dict(tot))


if __name__ == "__main__":
    main()
