#!/bin/bash
# Run on the GPU box: LDS / SQ counters of the full loop's dL/dparam kernels (are the LDS atomics of ray-coherent samples conflict bound?)
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_fll && rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/prof_fll -o p -- python $ROOT/tools/prof_full_loop.py > /tmp/fll.log 2>&1
python $ROOT/tools/prof_summary.py "$(find /tmp/prof_fll -name '*.db' | head -1)" pmc 2>&1 | grep -E "counter|k_pair" > $ROOT/gpurun_out/${TAG}_full_loop_lds_counters.txt
cat $ROOT/gpurun_out/${TAG}_full_loop_lds_counters.txt | cut -c1-160
