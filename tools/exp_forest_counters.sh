#!/bin/bash
# Run on the GPU box: what bounds the forest LoTD kernels (extra.forest_lotd: 8 blocks x 8 levels, 2^20 points) -- L2
# requests, hits / misses and the reads that leave the L2 (TCC_EA0_RDREQ, 32 / 64-byte requests towards the fabric).
# usage: tools/exp_forest_counters.sh <tag>
set -u
TAG=${1:-forest}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python -c \"import sys; sys.path.insert(0, '$ROOT'); import torch, bench; print(bench.forest_lotd_rate(torch.device('cuda', 0), iters=3))\""
: > "$OUT/${TAG}_forest_counters.txt"
rm -rf /tmp/prof_f && eval rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o p -- $CMD > /tmp/f.log 2>&1
python $ROOT/tools/prof_summary.py "$(find /tmp/prof_f -name '*.db' | head -1)" | head -12 | cut -c1-150 >> "$OUT/${TAG}_forest_counters.txt"
for CTRS in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" "TCC_BUSY_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum"; do
  rm -rf /tmp/prof_fc && eval rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/prof_fc -o p -- $CMD > /tmp/fc.log 2>&1
  DB=$(find /tmp/prof_fc -name '*.db' | head -1)
  if [ -z "$DB" ]; then echo "pass failed: $CTRS" >> "$OUT/${TAG}_forest_counters.txt"; tail -2 /tmp/fc.log >> "$OUT/${TAG}_forest_counters.txt"; continue; fi
  python $ROOT/tools/prof_summary.py "$DB" pmc | grep -E "counter|k_forest|k_bin_forest|k_accum" | cut -c1-150 >> "$OUT/${TAG}_forest_counters.txt"
  echo >> "$OUT/${TAG}_forest_counters.txt"
done
cat "$OUT/${TAG}_forest_counters.txt"
