#!/bin/bash
# Run on the GPU box: PMC counters of the fp32 decoder kernels (one shape, forward + backward) -> gpurun_out/<tag>_mlp_counters.txt
# usage: tools/gpu_mlp_counters.sh <tag> "32,64,64,16" [float|half]
TAG=${1:-mlpctr}; DIMS=${2:-32,64,64,16}; DT=${3:-float}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${TAG}_mlp_${3:-float}_counters.txt; mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
cat > /tmp/mlp_run.py <<PY
import sys, torch
sys.path.insert(0, "$ROOT")
from nr3d_lib_amd.models.blocks import MLP
dev = torch.device("cuda:0"); dims = [$DIMS]; n = 1 << 22
torch.manual_seed(0)
net = MLP(dims[0], dims[-1], D=len(dims) - 2, W=dims[1], dtype=torch.$DT, device=dev)
x = torch.randn(n, dims[0], device=dev, dtype=torch.$DT); gy = torch.randn(n, dims[-1], device=dev, dtype=torch.$DT)
for _ in range(3):
    xr = x.detach().requires_grad_(True); net.zero_grad(set_to_none=True); net(xr).backward(gy)
torch.cuda.synchronize()
PY
: > $OUT
rocprofv3 -L 2>/dev/null | grep -oE "(SQC_ICACHE|SQ_INSTS_MFMA|SQ_VALU_MFMA|SQ_INST_CYCLES|SQ_IFETCH|SQ_WAIT_INST|SQ_ACTIVE_INST|SQ_INSTS_VALU|SQ_BUSY_CY|SQ_WAVE_CY)[A-Z_0-9]*" | sort -u | tr '\n' ' ' >> $OUT; echo >> $OUT
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
            "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_ANY" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"; do
  rm -rf /tmp/prof_mc && timeout 300 rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/prof_mc -o p -- python /tmp/mlp_run.py > /tmp/mc.log 2>&1
  DB=$(find /tmp/prof_mc -name '*.db' | head -1)
  if [ -z "$DB" ]; then echo "pass failed: $CTRS" >> $OUT; tail -3 /tmp/mc.log >> $OUT; continue; fi
  python $ROOT/tools/prof_summary.py "$DB" pmc | grep -E "counter|k_mlp" | cut -c1-170 >> $OUT
  echo >> $OUT
done
cat $OUT
