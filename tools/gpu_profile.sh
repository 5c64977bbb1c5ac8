#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace (+ optional PMC passes) of bench.py, summarised to text under gpurun_out/<tag>/.
# usage: tools/gpu_profile.sh <tag> [pmc]       (databases stay in /tmp: gpurun_out is size-limited)
set -u
TAG=${1:-prof}; PMC=${2:-}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra"   # the timed loop only
rm -rf /tmp/prof_k && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_k -o p -- $BENCH > "$OUT/bench_under_rocprof.log" 2>&1
DB=$(find /tmp/prof_k -name '*.db' | head -1)
python $ROOT/tools/prof_summary.py "$DB" > "$OUT/bench_kernel_stats.txt" 2>&1
# second pass: the extra figures only (march + composite at 4096 / 262144 rays, full loop); headline loop reduced to 1 step
rm -rf /tmp/prof_x && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o p -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $ROOT/tools/prof_summary.py "$(find /tmp/prof_x -name '*.db' | head -1)" > "$OUT/bench_extra_kernel_stats.txt" 2>&1
if [ -n "$PMC" ]; then
  rm -f "$OUT/pmc_traffic.json"
  for CTR in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/prof_c && timeout 300 rocprofv3 --kernel-trace --pmc $CTR -d /tmp/prof_c -o p -- $BENCH > /dev/null 2>&1
    DB=$(find /tmp/prof_c -name '*.db' | head -1)
    python $ROOT/tools/prof_summary.py "$DB" pmc > "$OUT/bench_pmc_$(echo $CTR | tr A-Z a-z).txt" 2>&1
    python $ROOT/tools/prof_summary.py "$DB" pmc-json "$OUT/pmc_traffic.json"
  done
fi
head -16 "$OUT/bench_kernel_stats.txt"
