# round 4: GEMM chain with the weight fragments staged through LDS too (k_mip_gemm_h<.., A_LDS = true>) against the shipped kernel
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04o; rm -rf $O; mkdir -p $O
for rep in 1 2; do for v in "" _alds; do for sc in 1 0; do
  echo "== gemm_h_bench$v 16384 200 $sc" >> $O/gemm.log
  timeout 300 python tools/run_with_telemetry.py tools/build/gemm_h_bench$v 16384 200 $sc 2>&1 | grep -v "amdgpu.ids\|per layer\|flags" >> $O/gemm.log
done; done; done
cat $O/gemm.log
