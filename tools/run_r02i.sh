bash tools/gpu_variants.sh r02i_pl "NR3D_LOTD_FWD_PAIRLANE=0,1" "NR3D_LOTD_LDS_STAGE=0,1"
python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-timers 2>&1 | tail -1 | cut -c1-300
python bench.py --steps 20 --warmup 5 > gpurun_out/r02i_bench.json 2> gpurun_out/r02i_bench.err; tail -3 gpurun_out/r02i_bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r02i_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['kernel_ms']); print(json.dumps(d['roofline'])[:1500]); print(json.dumps(d['extra']['march_composite'])[:1500]); print({k:(v.get('ms_per_iter') or v.get('ms_total') or v.get('ms_per_step')) for k,v in d['extra'].items()}); print(d.get('cpu_baseline'))"
