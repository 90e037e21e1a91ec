# round 6: counters of the fused training-chain kernels (k_tp_chain_fwd / k_tp_chain_bwd) on one fine level (tools/bench_train_chain.py),
# the pix / chain tests after the view-mean fold in the PixelNeRF chain, the training-step line and its kernel stats
cd $GRAFT_REPO_ROOT; T=${1:-r06w}; O=gpurun_out/$T; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_pix_training.py tests/test_gpu_host_r6.py tests/test_gpu_pixelnerf.py -q -m gpu -x > $O/pytest_a.log 2>&1; echo "pytest rc=$?" >> $O/pytest_a.log
tail -3 $O/pytest_a.log
bash tools/pmc_kernel.sh pmc_train_chain_fwd tools/bench_train_chain.py "k_tp_chain_fwd" > $O/pmc_fwd.log 2>&1
cp gpurun_out/pmc_train_chain_fwd/summary.json $O/pmc_train_chain_fwd.json
python tools/pmc_summarize.py gpurun_out/pmc_train_chain_fwd "k_tp_chain_bwd" --note "same passes, backward kernel" > $O/pmc_train_chain_bwd.json 2>$O/pmc_bwd.err
python tools/pmc_summarize.py gpurun_out/pmc_train_chain_fwd "k_dw<128>" --note "same passes, weight-gradient GEMM" > $O/pmc_train_dw.json 2>>$O/pmc_bwd.err
timeout 150 python bench.py --workload neo360_train --steps 10 --warmup 3 > $O/train.json 2> $O/train.err
python - <<PY
import json
t = json.load(open("$O/train.json")); print("train: ms/step %.2f rays/s %.0f" % (t["ms_per_step"], t["value"]), t.get("phases_ms"), t["roofline"]["frac"])
for f in ("pmc_train_chain_fwd", "pmc_train_chain_bwd", "pmc_train_dw"):
    try:
        d = json.load(open("$O/%s.json" % f)); print(f, {k: d.get(k) for k in ("avg_ms", "launches", "mfma_busy", "valu_per_mfma", "lds_conflict_frac", "l2_hit", "hbm_bytes_per_launch", "tcp_tcc_read_bytes_per_launch")})
    except Exception as e: print(f, "?", e)
PY
timeout 200 python tools/bench_train_other.py > $O/train_other.log 2>&1; tail -5 $O/train_other.log
