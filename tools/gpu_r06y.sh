cd $GRAFT_REPO_ROOT; O=gpurun_out/r06y; rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_host_r6.py -q -m gpu -x -k "shared_gradient or merged" 2>&1 | grep -v amdgpu | tee $O/pytest.log | tail -15
