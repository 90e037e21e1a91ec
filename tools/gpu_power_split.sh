# which operands carry the power of k_tp_mlp_hp: the same instruction stream with (weights, features) = real / zero in the four
# combinations, rocm-smi sampled while it runs.  usage: bash tools/gpu_power_split.sh [slot]
S=${1:-1}; D=gpurun_out/r03p; mkdir -p $D; L=$D/power_split_slot$S.log; : > $L
sample() { for i in $(seq 3); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Current Socket Graphics Package Power" | tr -s ' \t' ' ' | tr '\n' ' '; echo; sleep 0.7; done; }
for combo in "1 1" "0 1" "1 0" "0 0"; do
  set -- $combo
  echo "== weights x$1, features x$2" >> $L
  ( sleep 7; sample ) >> $L &
  SP=$!
  SCALE_W=$1 SCALE_F=$2 POLL=0 PREC=f16x3 R=8192 N=385 SLOT=$S REPS=1600 python tools/bench_tp_kernel.py 2>/dev/null | tail -1 >> $L
  wait $SP
done
cat $L
