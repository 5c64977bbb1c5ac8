#!/bin/bash
# Run on the GPU box: PMC counters of the sorted-points dL/dparam kernels on the reference's forest workload (one counter group per pass,
# kernel trace only) -> gpurun_out/<tag>/forest_sorted_counters.txt.   usage: tools/gpu_forest_counters.sh <tag>
set -u
TAG=${1:-forestctr}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
bash $ROOT/tools/gpu_forest_prof.sh ${TAG}_trace 3 > /dev/null 2>&1        # leaves /tmp/forest_run.py behind
cd /tmp && export TMPDIR=/tmp
: > "$OUT/forest_sorted_counters.txt"
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
            "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU" \
            "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  rm -rf /tmp/prof_fc && timeout 300 rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/prof_fc -o p -- python /tmp/forest_run.py > /tmp/fc.log 2>&1
  DB=$(find /tmp/prof_fc -name '*.db' | head -1)
  if [ -z "$DB" ]; then echo "pass failed: $CTRS" >> "$OUT/forest_sorted_counters.txt"; tail -2 /tmp/fc.log >> "$OUT/forest_sorted_counters.txt"; continue; fi
  python $ROOT/tools/prof_summary.py "$DB" pmc | grep -E "counter|k_vm_sorted|k_vs_" | cut -c1-150 >> "$OUT/forest_sorted_counters.txt"
  echo >> "$OUT/forest_sorted_counters.txt"
done
cat "$OUT/forest_sorted_counters.txt"
