#!/bin/bash
# Run on the GPU box: rocprofv3 kernel trace of the full loop alone (bench.py full_loop_rate) -> gpurun_out/<tag>_full_loop_kernel_stats.txt
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_fl && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_fl -o p -- python $ROOT/tools/prof_full_loop.py > /tmp/fl.log 2>&1
grep "^{" /tmp/fl.log | tail -1
python $ROOT/tools/prof_summary.py "$(find /tmp/prof_fl -name '*.db' | head -1)" > $ROOT/gpurun_out/${TAG}_full_loop_kernel_stats.txt 2>&1
head -36 $ROOT/gpurun_out/${TAG}_full_loop_kernel_stats.txt | cut -c1-150
