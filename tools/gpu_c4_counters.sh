#!/bin/bash
# Run on the GPU box: kernel statistics + SQ / LDS / TCC counters of configs[3] (tools/bench_c4.py) -> gpurun_out/<tag>_c4_*.txt
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/bench_c4.py --iters 2"
rm -rf /tmp/prof_c4 && rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o p -- $CMD > /tmp/c4.log 2>&1
python $ROOT/tools/prof_summary.py "$(find /tmp/prof_c4 -name '*.db' | head -1)" > $ROOT/gpurun_out/${TAG}_c4_kernel_stats.txt 2>&1
: > $ROOT/gpurun_out/${TAG}_c4_counters.txt
for CTRS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_BUSY_sum"; do
  rm -rf /tmp/prof_c4c && rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/prof_c4c -o p -- $CMD > /tmp/c4c.log 2>&1
  python $ROOT/tools/prof_summary.py "$(find /tmp/prof_c4c -name '*.db' | head -1)" pmc 2>&1 | grep -E "counter|k_bin|k_accum|k_fwd|k_bwd_bwd|k_cp_direct" >> $ROOT/gpurun_out/${TAG}_c4_counters.txt
  echo >> $ROOT/gpurun_out/${TAG}_c4_counters.txt
done
head -20 $ROOT/gpurun_out/${TAG}_c4_kernel_stats.txt | cut -c1-150
cut -c1-150 $ROOT/gpurun_out/${TAG}_c4_counters.txt
