# repeat of the full GPU suite on another box (flakiness check of the round's new tests)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05y; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python -m pytest tests/test_gpu_mip_training.py tests/test_gpu_pix_training.py tests/test_gpu_training.py -q -m gpu > $O/pytest2.log 2>&1; tail -2 $O/pytest2.log
