# round 5: frame-level A/B of the dense evaluators with JOULES PER FRAME (VERDICT r4 tasks 3 / 4: "... or a measured negative with joules
# per frame"): vanilla 64-point tile (product) / 128-point tile = two M-tiles per weight fetch / 8 waves / 8-byte epilogue stores /
# 3-instruction split; Mip-NeRF 360 64 + 128 with the 3-instruction split in the GEMM epilogues
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05h; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
B=$PWD/tools/build
frame() { # tag workload lib [env]
  env $4 NEO360_HIP_LIB=$3 timeout 300 python bench.py --workload $2 --steps 8 --warmup 2 --cpu-rays 0 --others 0 --exact-f32 0 > $O/$2_$1.json 2> $O/$2_$1.err
  python - "$1" "$2" <<'PY'
import json, sys
tag, wl = sys.argv[1], sys.argv[2]
try:
    d = json.load(open("gpurun_out/r05h/%s_%s.json" % (wl, tag))); r = d["roofline"]
    print("%-11s %-12s rays/s %9.0f  ms/frame %7.1f  sclk %4.0f MHz  power %4.0f W  JOULES/FRAME %6.1f  alg TFLOP/s %.1f (%.1f %% of 833)" % (
        wl, tag, d["value"], d["ms_per_step"], r["sclk_mhz_mean"], r["power_w_mean"], d["ms_per_step"] * 1e-3 * r["power_w_mean"], r["achieved"], 100 * r["achieved"] / 833.3))
except Exception as e:
    print(wl, tag, "failed", e)
PY
}
frame tile64      vanilla ""                      NEO_X=0 | tee -a $O/dense_ab.log
frame tile128     vanilla ""                      NEO_VANILLA_H_TILE=128 | tee -a $O/dense_ab.log
frame waves8      vanilla ""                      NEO_VANILLA_H_WAVES=8 | tee -a $O/dense_ab.log
frame store64     vanilla $B/libneo_vh_st0.so     NEO_X=0 | tee -a $O/dense_ab.log
frame mixlo       vanilla $B/libneo_vh_mixlo.so   NEO_X=0 | tee -a $O/dense_ab.log
frame tile64b     vanilla ""                      NEO_X=0 | tee -a $O/dense_ab.log
frame product     mip360_128 ""                   NEO_X=0 | tee -a $O/dense_ab.log
frame mixlo       mip360_128 $B/libneo_mip_mixlo.so NEO_X=0 | tee -a $O/dense_ab.log
frame product2    mip360_128 ""                   NEO_X=0 | tee -a $O/dense_ab.log
