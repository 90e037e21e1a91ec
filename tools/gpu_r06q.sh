# round 6: the fused PixelNeRF / Mip-NeRF 360 training chains: their tests, then step times fused vs per-layer (module.train_fused = False)
cd $GRAFT_REPO_ROOT; T=${1:-r06q}; O=gpurun_out/$T; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pix_training.py tests/test_gpu_mip_training.py tests/test_abi.py -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python tools/bench_train_other.py > $O/train_other.log 2>&1
tail -15 $O/pytest.log; cat $O/train_other.log | tail -8
