# round 5: Mip-NeRF 360 training call on the operator chain
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05s; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_mip_training.py tests/test_gpu_mip360.py -q -m gpu --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -60 $O/pytest.log
