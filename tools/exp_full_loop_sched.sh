#!/bin/bash
# Needs the experiments build (round 4): make -C nr3d_lib_amd/csrc clean && make -C nr3d_lib_amd/csrc -j8 EXTRA=-DNR3D_EXPERIMENTS
# Run on the GPU box: the full loop (bench.full_loop_rate) under the forward's block schedules
# (NR3D_LOTD_SCHED: 3 = cost-balanced XCD-affine work line (default), 1 = level q -> XCD q % 8, 2 = chunk-major, no affinity)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for m in 3 1 2 0; do
  echo "NR3D_LOTD_SCHED=$m: $(NR3D_LOTD_SCHED=$m python $ROOT/tools/prof_full_loop.py 2>/dev/null | grep '^{' | tail -1)"
done
