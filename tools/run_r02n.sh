#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
: > gpurun_out/r02n_chunk.txt
for lg in 16 17 18 19 20; do
  NR3D_PAIR_ALL=0 NR3D_LOTD_BIN_CHUNK_LOG2=$lg python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pk=d['roofline']['per_kernel']
print($lg, d['ms_per_step'], d['kernel_ms'], {k:(round(v['avg_us']*v['launches_per_step'],1), v['launches_per_step']) for k,v in pk.items() if 'pair' in k})" | tee -a gpurun_out/r02n_chunk.txt
done
