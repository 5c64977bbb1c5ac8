#!/bin/bash
# Run on the GPU box: L2 request counters of the full loop's kernels -> gpurun_out/<tag>_full_loop_counters.txt
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_flc && rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d /tmp/prof_flc -o p -- python $ROOT/tools/prof_full_loop.py > /tmp/flc.log 2>&1
python $ROOT/tools/prof_summary.py "$(find /tmp/prof_flc -name '*.db' | head -1)" pmc 2>&1 | grep -E "counter|k_fwd_pairlane|k_pair|k_fwd_lds" > $ROOT/gpurun_out/${TAG}_full_loop_counters.txt
cat $ROOT/gpurun_out/${TAG}_full_loop_counters.txt | cut -c1-160
