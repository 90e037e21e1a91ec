# round 5, first GPU call: the whole -m gpu suite on the round-5 defaults (doubling-order encodings, mixlo split, chained plane
# blends, second bias read; new host paths), then the evaluator A/B on one box:
#   r4 = round-4 kernels   pair2 / mixlo / fmab = one change each   default = all   all_dpf0 / all_xs1 = all with fewer registers
# usage: bash tools/gpu_r05a.sh
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05a; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
./tools/build/split2_check_mixlo > $O/split2_check.log 2>&1; echo "split2_check rc=$?" >> $O/split2_check.log; tail -3 $O/split2_check.log
timeout 1200 python -m pytest tests -q -m gpu --durations=12 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
K=$O/kernel_ab.log
run() { # tag lib
  NEO360_HIP_LIB=$2 PP=3 TAG=$1 SLOTS=1,3,0,2 REPS=40 timeout 300 python tools/bench_tp_kernel.py 2>&1 | grep -E "slot|Error|error" >> $K; }
B=$PWD/tools/build
run r4       $B/libneo_r4.so
run pair2    $B/libneo_pair2.so
run mixlo    $B/libneo_mixlo.so
run fmab     $B/libneo_fmab.so
run default  ""
run all_dpf0 $B/libneo_all_dpf0.so
run all_xs1  $B/libneo_all_xs1.so
run nopair2  $B/libneo_nopair2.so
run r4       $B/libneo_r4.so
run default  ""
cat $K
python - <<'PY'
import re
rows = {}
for l in open("gpurun_out/r05a/kernel_ab.log"):
    m = re.match(r"(\S+) f16x3 pp=\d slot (\d) .*?([\d.]+) ms .*?sclk (\S+) MHz\s+power (\S+) W", l)
    if m:
        tag, slot, ms, clk, pw = m.group(1), int(m.group(2)), float(m.group(3)), m.group(4), m.group(5)
        rows.setdefault(tag, []).append((slot, ms, clk, pw))
print("%-10s %8s %8s %8s %8s | %8s  %s" % ("variant", "in-fine", "out-fine", "in-crs", "out-crs", "sum ms", "joules per launch (W x ms), clocks"))
for tag, r in rows.items():
    for i in range(0, len(r), 4):
        g = {s: (ms, clk, pw) for s, ms, clk, pw in r[i:i + 4]}
        if len(g) < 4:
            continue
        j = ["%.2f" % (float(g[s][2]) * g[s][0] * 1e-3) if g[s][2] != "?" else "?" for s in (1, 3, 0, 2)]
        print("%-10s %8.3f %8.3f %8.3f %8.3f | %8.3f  J %s  MHz %s" % (tag, g[1][0], g[3][0], g[0][0], g[2][0], sum(g[s][0] for s in g),
              "/".join(j), "/".join(g[s][1] for s in (1, 3, 0, 2))))
PY
timeout 600 python bench.py --steps 8 --warmup 2 --cpu-rays 0 --others 0 --exact-f32 0 > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json; tail -3 $O/bench.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05a/bench.json")); r = d["roofline"]
    print("rays/s %.0f" % d["value"], "ms %.1f" % d["ms_per_step"], "kernel", r["kernel"], "avg launch ms %.2f" % r["avg_launch_ms"], "sclk", r.get("sclk_mhz_mean"),
          "power", r.get("power_w_mean"), "frac %.4f" % r["frac"], "scene_setup", d.get("scene_setup", {}).get("runs_ms"), {k: round(v["avg_launch_ms"], 2) for k, v in r["kernels"].items()})
except Exception as e:
    print("bench failed", e)
PY
