#!/bin/bash
# Run on the GPU box: headline loop under combinations of environment knobs; one line per variant (ms_per_step, kernel_ms).
# usage: tools/gpu_variants.sh <tag> "VAR1=a,b,c" "VAR2=x,y" ...   (cartesian product)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG.txt; : > "$OUT"
combos=("")
for spec in "$@"; do
  name=${spec%%=*}; IFS=',' read -ra vals <<< "${spec#*=}"
  next=()
  for c in "${combos[@]}"; do for v in "${vals[@]}"; do next+=("$c $name=$v"); done; done
  combos=("${next[@]}")
done
for c in "${combos[@]}"; do
  line=$(env $c python $ROOT/bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms'])" 2>/dev/null)
  echo "$c : $line" | tee -a "$OUT"
done
