#!/bin/bash
# Run on the GPU box: several rocprofv3 --pmc passes of the headline loop (bench.py --no-extra), one counter group per
# pass (SQ 8 / TCC 4 / TCP 4 slots; never combined with sys/hip traces), per-kernel averages appended to
# gpurun_out/<tag>/counters.txt.
# usage: tools/gpu_counters.sh <tag> [kernel-regex]
set -u
TAG=${1:-ctr}; KRE=${2:-k_}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
PASSES=(
 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
 "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum"
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_LEVEL_WAVES GRBM_GUI_ACTIVE"
 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU"
 "TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
 "TCC_BUSY_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
)
: > "$OUT/counters.txt"
for CTRS in "${PASSES[@]}"; do
  rm -rf /tmp/prof_c
  timeout 300 rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/prof_c -o p -- $BENCH > /tmp/pmc.log 2>&1
  DB=$(find /tmp/prof_c -name '*.db' | head -1)
  if [ -z "$DB" ]; then echo "pass failed: $CTRS" >> "$OUT/counters.txt"; tail -3 /tmp/pmc.log >> "$OUT/counters.txt"; continue; fi
  python $ROOT/tools/prof_summary.py "$DB" pmc | grep -E "counter|$KRE" >> "$OUT/counters.txt"
  echo >> "$OUT/counters.txt"
done
cat "$OUT/counters.txt"
