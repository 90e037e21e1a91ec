mkdir -p gpurun_out/r02m; cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu --durations=10 > gpurun_out/r02m/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02m/pytest.log
python tools/bench_pillar.py > gpurun_out/r02m/pillar.log 2>&1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r02m/prof_pillar -o pillar -- python tools/bench_pillar.py > gpurun_out/r02m/prof_pillar.log 2>&1
tail -4 gpurun_out/r02m/pytest.log; cat gpurun_out/r02m/pillar.log | tail -2
