#!/bin/bash
# counter passes over the training GEMM (k_sgemm) inside the vanilla NeRFMLP forward (256-wide layers, aligned rows)
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_sgemm; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/sgemm_probe.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
from neo360_amd import models, synth, training
dev = "cuda"
vm = models.NeRFMLP().to(dev)
sd = synth.vanilla_state(0)
vm.load_state_dict({k[len("fine_mlp."):]: v for k, v in sd.items() if k.startswith("fine_mlp.")})
x, d = torch.randn(4096, 193, 63, device=dev), torch.randn(4096, 27, device=dev)
with torch.no_grad():
    for _ in range(2):
        training.nerf_mlp(vm, x, d)
torch.cuda.synchronize()
PY
i=0
while read -r P; do
  [ -z "$P" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P -f csv -d $OUT/pass$i -- python /tmp/sgemm_probe.py $REPO > $OUT/pass$i.log 2>&1
done <<'PASSES'
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
PASSES
python $REPO/tools/pmc_summarize.py $OUT "k_sgemm<false, false>" > $OUT/summary.json
cat $OUT/summary.json
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*agent_info.csv" -delete
