# power / clock envelope of the split-fp16 kernels: same instruction stream with random and with all-zero operands
# usage: gpu_power.sh vanilla|neo360
D=${POWER_OUT:-gpurun_out/r02l}; mkdir -p $D; L=$D/power_$1.log; : > $L
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" >> $L
sample() { for i in $(seq 8); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Current Socket Graphics Package Power" | tr -s ' \t' ' ' | tr '\n' ' '; echo; sleep 0.5; done; }
for mode in random zero; do
  echo "== $1, $mode operands" >> $L
  if [ $mode = zero ]; then export SCALE=0; else export SCALE=1; fi
  ( sleep 5; sample ) >> $L &
  SP=$!
  if [ $1 = vanilla ]; then PREC=f16x3 R=65536 N=193 REPS=200 TAG=$mode python tools/bench_kernel.py 2>/dev/null >> $L
  else PREC=f16x3 R=8192 N=385 SLOT=1 REPS=600 TAG=$mode python tools/bench_tp_kernel.py 2>/dev/null >> $L; fi
  wait $SP
done
cat $L
