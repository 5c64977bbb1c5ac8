#!/bin/bash
# usage: tools/gpu_pmc.sh <tag> "<CTR1 CTR2 ...>" [kernel-regex]   -- one rocprofv3 --pmc pass of bench.py, per-kernel averages
set -u
TAG=$1; CTRS=$2; KRE=${3:-k_}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c && rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/prof_c -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pmc.log 2>&1
DB=$(find /tmp/prof_c -name '*.db' | head -1)
if [ -z "$DB" ]; then tail -5 /tmp/pmc.log; exit 1; fi
python $ROOT/tools/prof_summary.py "$DB" pmc | grep -E "counter|$KRE" | tee -a "$OUT/pmc_$(echo $CTRS | tr ' ' '_' | cut -c1-60).txt"
