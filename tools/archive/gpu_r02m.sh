# the round's closing GPU call: full GPU test suite, smoke, default bench line, rocprofv3 kernel stats of the same command
mkdir -p gpurun_out/r02z; cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r02z/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02z/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02z/smoke.log 2>&1
python bench.py > gpurun_out/r02z/bench.json 2> gpurun_out/r02z/bench.err
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/r02z/prof -o bench -- python bench.py --steps 2 --cpu-rays 0 --others 0 > gpurun_out/r02z/prof.log 2>&1
find gpurun_out/r02z/prof -name "*kernel_trace.csv" -delete; find gpurun_out/r02z/prof -name "*agent_info.csv" -delete
tail -4 gpurun_out/r02z/pytest.log; cat gpurun_out/r02z/smoke.log | tail -2; cut -c1-200 gpurun_out/r02z/bench.json
