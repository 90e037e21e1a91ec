# round 5: PixelNeRF training call on the operator chain (neo_linear_*, neo_pix_gather_map)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05r; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pix_training.py tests/test_gpu_pixelnerf.py tests/test_gpu_host_r5.py -q -m gpu -x --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
