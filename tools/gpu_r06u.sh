# round 6: where the fused training chain's time goes: NEO_CHAIN_ABLATE variants of train_mlp.hip on the chain op alone (577,500 rows)
cd $GRAFT_REPO_ROOT; T=${1:-r06u}; O=gpurun_out/$T; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for v in 1 2 4 3 7; do python tools/build_variant.py abl$v train_mlp.hip -DNEO_CHAIN_ABLATE=$v > /dev/null 2>&1; done
{
for rep in 1 2; do
  TAG=base timeout 100 python tools/bench_train_chain.py
  for v in 1 2 4 3 7; do TAG=ablate$v NEO360_HIP_LIB=tools/build/libneo_abl$v.so timeout 100 python tools/bench_train_chain.py; done
done
TAG=layers NEO360_TRAIN_CHAIN=0 timeout 100 python tools/bench_train_chain.py
TAG=base_ch4 CH=4 timeout 100 python tools/bench_train_chain.py
} 2>&1 | grep -v amdgpu.ids | tee $O/ablate.log
