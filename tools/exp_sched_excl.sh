#!/bin/bash
# Needs the experiments build (round 4): make -C nr3d_lib_amd/csrc clean && make -C nr3d_lib_amd/csrc -j8 EXTRA=-DNR3D_EXPERIMENTS
# Run on the GPU box: the two-lane forward's schedule -- work line (NR3D_LOTD_SCHED_EXCL=0) vs exclusive fine levels + shared
# coarse levels (1, default): headline loop at 2^20 / 2^22 and the full loop
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for m in 0 1; do
  for n in 20 22; do
  echo "NR3D_LOTD_SCHED_EXCL=$m 2^$n: $(NR3D_LOTD_SCHED_EXCL=$m python $ROOT/bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --log2-points $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms'], d['roofline']['avg_launch_us'])")"
  done
  echo "NR3D_LOTD_SCHED_EXCL=$m full loop: $(NR3D_LOTD_SCHED_EXCL=$m python $ROOT/tools/prof_full_loop.py 2>/dev/null | grep '^{' | tail -1 | cut -c120-400)"
done
