# round 6: 128 x 128 tiles for the big training GEMMs (texel-space projection, Mip trunk): A/B of the step lines
cd $GRAFT_REPO_ROOT; T=${1:-r06x}; O=gpurun_out/$T; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_host_r5.py tests/test_gpu_pix_training.py tests/test_gpu_mip_training.py -q -m gpu -x -k "linear or project or fused or gradients" > $O/pytest_a.log 2>&1; echo "pytest rc=$?" >> $O/pytest_a.log
tail -3 $O/pytest_a.log
for w in 1 0 1 0; do
  NEO360_WIDE_TILES=$w timeout 150 python bench.py --workload neo360_train --steps 10 --warmup 3 --cpu-rays 0 > $O/train_$w.json 2> $O/train_$w.err || { tail -3 $O/train_$w.err; continue; }
  python - <<PY
import json
t = json.load(open("$O/train_$w.json")); print("wide=$w train: ms/step %.2f rays/s %.0f" % (t["ms_per_step"], t["value"]), t.get("phases_ms"))
PY
done
for w in 1 0; do echo "wide=$w"; NEO360_WIDE_TILES=$w timeout 200 python tools/bench_train_other.py 2>&1 | grep -v amdgpu | tee -a $O/train_other_$w.log; done
NEO360_WIDE_TILES=1 timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_train -o train -- python bench.py --workload neo360_train --steps 5 --warmup 2 --cpu-rays 0 > $O/prof_train.log 2>&1
find $O/prof_train -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_train.csv \;
find $O/prof_train -name "*kernel_trace.csv" -delete; find $O/prof_train -name "*agent_info.csv" -delete
grep "k_sgemm" $O/kernel_stats_train.csv | cut -c1-60,170-260
