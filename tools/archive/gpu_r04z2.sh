# round 4's closing GPU call, second edition (after the layer-by-layer Mip-NeRF 360 NeRF MLP): everything tools/gpu_r04z.sh does,
# plus counter passes and kernel stats of the Mip-NeRF 360 workload and the GEMM-chain micro-benchmark.
# usage: bash tools/gpu_r04z2.sh [tag]
cd $GRAFT_REPO_ROOT; T=${1:-r04z2}
bash tools/gpu_r04z.sh $T > gpurun_out/$T.stdout 2>&1
O=gpurun_out/$T
export TMPDIR=/tmp
bash tools/pmc_bench.sh mip360 f16x3 "k_mip_gemm_h" "k_mip_mlp_h<1024" > $O/pmc_bench_mip.log 2>&1
cp gpurun_out/pmc_mip360_f16x3/summary.json profiles/r04_pmc_mip360_f16x3.json
cp gpurun_out/pmc_mip360_f16x3/summary_2.json profiles/r04_pmc_mip360_f16x3_tail.json
cp gpurun_out/pmc_mip360_f16x3/kernel_stats.csv $O/kernel_stats_mip360.csv
for args in "16384 200 1" "16384 200 0"; do
  echo "== gemm_h_bench $args" >> $O/gemm_chain.log
  timeout 300 python tools/run_with_telemetry.py tools/build/gemm_h_bench $args 2>&1 | grep -v amdgpu.ids >> $O/gemm_chain.log
done
for wl in mip360_128 mip360; do for lay in 0 1; do
NEO360_MIP_LAYERED=$lay timeout 600 python bench.py --workload $wl --steps 3 --warmup 1 --cpu-rays 0 --others 0 --exact-f32 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$wl layered=$lay rays/s %.0f ms %.1f' % (d['value'], d['ms_per_step']), '|', r.get('kernel'), 'avg launch %.2f ms' % r['avg_launch_ms'], 'alg TFLOP/s %.1f' % r['achieved'], 'frac split %.3f' % (r.get('frac_of_split_ceiling') or 0), 'mfma_busy', r.get('mfma_busy'), round(r.get('sclk_mhz_mean') or 0), 'MHz', round(r.get('power_w_mean') or 0), 'W')
for k, v in r['kernels'].items(): print('    ', k, v['launches'], '%.3f ms' % v['avg_launch_ms'], '%.1f TF' % v['algorithmic_tflops'])" >> $O/mip_bench.log
done; done
find gpurun_out -name "*counter_collection.csv" -size +200k -delete
find gpurun_out -name "*kernel_trace.csv" -size +200k -delete
tail -40 gpurun_out/$T.stdout; cat $O/gemm_chain.log $O/mip_bench.log; cat gpurun_out/pmc_mip360_f16x3/summary.json | head -60
