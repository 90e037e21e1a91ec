# round 4, task 8: the Mip-NeRF 360 trunk as a layer-by-layer split-fp16 GEMM (csrc/mip_gemm_h.h) - indexing check and
# sustained rate (tools/gemm_h_bench.hip) with clock / power; usage: gpu_r04h.sh [binary suffixes...]
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04h; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for v in "" "$@"; do
  B=tools/build/gemm_h_bench$v
  for args in "16384 200 1" "16384 200 0"; do
    echo "== gemm_h_bench$v $args" >> $O/gemm.log
    timeout 300 python tools/run_with_telemetry.py $B $args 2>&1 | grep -v amdgpu.ids >> $O/gemm.log
  done
done
cat $O/gemm.log
