for c in 22 23 24; do for u in 1536 3072; do
  echo -n "chunk=$c units=$u : "; NR3D_LOTD_BIN_CHUNK_LOG2=$c NR3D_PAIR_UNITS=$u python bench.py --log2-points 24 --steps 5 --warmup 2 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms'], d['roofline']['whole_step_frac'])"
done; done
