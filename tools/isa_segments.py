"""Static anatomy of a kernel's ISA: instruction classes between consecutive s_barrier instructions and per basic block.

    hipcc ... --cuda-device-only -S mlp_tp_hp.hip -o tp_hp.s ;  python tools/isa_segments.py tp_hp.s k_tp_mlp_hpILi3E

Prints, in program order, every basic block (label, backward-branch targets marked as loops) with its counts of MFMA, other
VALU, LDS, vector-memory, scalar and waitcnt instructions, and a running segment number that increases at each s_barrier.
Used for profiles/r03_tp_hp_isa_anatomy.log: where the VALU instructions of a tile sit."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ": " in l and "@" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks, cur, label = [], collections.Counter(), "entry"
    seg = 0
    order = {}
    ops = collections.Counter()
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        if not t or t.startswith((";", "//", ".")) and not re.match(r"^\.LBB\S*:", t):
            continue
        m = re.match(r"^(\.LBB\S*):", t)
        if m:
            blocks.append([label, cur, seg, None])
            label, cur = m.group(1), collections.Counter()
            order[label] = len(blocks)
            continue
        op = t.split()[0]
        c = classify(op)
        cur[c] += 1
        if c == "valu":
            ops[(seg, op)] += 1
        if c == "branch":
            cur["->" + t.split()[-1]] += 1
        if c == "barrier":
            blocks.append([label, cur, seg, None])
            seg += 1
            label, cur = label + "'", collections.Counter()
    blocks.append([label, cur, seg, None])
    print("%-14s %4s %5s %5s %4s %5s %5s %5s %4s  branches" % ("block", "seg", "mfma", "valu", "lds", "vmem", "salu", "wait", "nop"))
    tot = collections.Counter()
    for i, (lab, c, sg, _) in enumerate(blocks):
        br = []
        for k in c:
            if k.startswith("->"):
                tgt = k[2:]
                br.append(tgt + (" (LOOP)" if order.get(tgt, 1 << 30) <= i else ""))
        n = sum(v for k, v in c.items() if not k.startswith("->"))
        if n == 0:
            continue
        for k, v in c.items():
            if not k.startswith("->"):
                tot[k] += v
        print("%-14s %4d %5d %5d %4d %5d %5d %5d %4d  %s" % (lab[:14], sg, c["mfma"], c["valu"], c["lds"], c["vmem"], c["salu"],
                                                            c["wait"], c["nop"], ", ".join(br)))
    print("static totals:", dict(tot))
    if len(sys.argv) > 3:
        want = int(sys.argv[3])
        print("VALU opcodes of segment", want)
        for (sg, op), n in sorted(ops.items(), key=lambda kv: -kv[1]):
            if sg == want:
                print("   %5d  %s" % (n, op))


if __name__ == "__main__":
    main()
