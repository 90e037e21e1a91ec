# round 6 (VERDICT r5 task 1a): the energy budget of one inside-sphere launch of k_tp_mlp_hp.
# NEO_TP_ABLATE variants (wrong results by construction: timing + energy probes only) x full frames with in-run telemetry;
# joules per launch = mean launch time of the kernel x mean socket power of the timed steps.  Idle socket power first.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06a; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
B=$PWD/tools/build
L=$O/energy_budget.log
echo "== idle socket power (nothing running), 8 samples" | tee -a $L
for i in $(seq 8); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Current Socket Graphics Package Power" | tr -s ' \t' ' ' | tr '\n' ' '; echo; sleep 0.5; done | tee -a $L
rocm-smi --showmaxpower 2>/dev/null | grep -i max | tee -a $L
frame() { # tag lib
  NEO360_HIP_LIB=$2 timeout 300 python bench.py --steps 8 --warmup 2 --cpu-rays 0 --others 0 --exact-f32 0 --setup-timing 0 > $O/frame_$1.json 2> $O/frame_$1.err
  python - "$1" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.load(open("gpurun_out/r06a/frame_%s.json" % tag)); r = d["roofline"]
    k = {n: round(v["avg_launch_ms"], 2) for n, v in r["kernels"].items()}
    j = {n: round(v["avg_launch_ms"] * 1e-3 * r["power_w_mean"], 1) for n, v in r["kernels"].items()}
    print("%-10s rays/s %8.0f  ms %.1f  %s  sclk %.0f  power %.0f W  joules/launch %s" % (tag, d["value"], d["ms_per_step"], k, r["sclk_mhz_mean"], r["power_w_mean"], j))
except Exception as e:
    print(tag, "failed", e)
PY
}
echo "== frames: tag = NEO_TP_ABLATE bits of mlp_tp_hp.hip (1 latent gathers, 2 plane gathers, 4 pos_enc, 8 streamed MFMAs, 16 L1..L3 GEMMs, 128 epilogue stores)" | tee -a $L
frame default ""  | tee -a $L
for a in 1 2 3 4 7 128 8 16 24 31; do frame ablate$a $B/libneo_ablate$a.so | tee -a $L; done
frame default2 "" | tee -a $L
echo "== isolated test launches of the inside-sphere MLPs (8192 rays x 385 fine / 129 coarse points), ~4 s each: the kernel's own power" | tee -a $L
micro() { # tag lib
  NEO360_HIP_LIB=$2 POLL=0 PREC=f16x3 R=8192 SLOTS=1,0 REPS=500 TAG=$1 timeout 200 python tools/bench_tp_kernel.py 2>/dev/null | tee -a $L
}
micro default ""
for a in 1 2 3 4 7 128 8 16 24 31; do micro ablate$a $B/libneo_ablate$a.so; done
micro default2 ""
cp $L gpurun_out/r06a_energy_budget.log
