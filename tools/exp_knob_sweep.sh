#!/bin/bash
# Run on the GPU box with the EXPERIMENTS build in place (make EXTRA=-DNR3D_EXPERIMENTS): same-box sweep of one NR3D_<knob> over values,
# headline step only.  usage: tools/exp_knob_sweep.sh KNOB v1 v2 ...   (the first value should be the default)
KNOB=$1; shift
for rep in 1 2; do
for v in "$@"; do
  echo -n "NR3D_$KNOB=$v: "
  env NR3D_$KNOB=$v python bench.py --no-extra --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('ms_per_step_median'), d['kernel_ms'])"
done; done
