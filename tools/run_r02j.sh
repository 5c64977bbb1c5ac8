python -m pytest tests/test_lotd_gpu.py tests/test_fullsize_gpu.py tests/test_pack_ops_gpu.py tests/test_ray_query_gpu.py -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r02j_pytest.log; tail -4 gpurun_out/r02j_pytest.log
bash tools/gpu_variants.sh r02j_pl "NR3D_LOTD_FWD_PAIRLANE=0,1"
python bench.py --steps 20 --warmup 5 > gpurun_out/r02j_bench.json 2> gpurun_out/r02j_bench.err; tail -3 gpurun_out/r02j_bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r02j_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['kernel_ms']); r=d['roofline']; print(r['kernel'], r['avg_launch_us'], r['frac'], r['whole_step_frac']); print({k:(v['avg_us']) for k,v in r['per_kernel'].items()}); print(json.dumps(d['extra']['march_composite'])[:900]); print({k:(v.get('ms_per_iter') or v.get('ms_total') or v.get('ms_per_step')) for k,v in d['extra'].items()}); print(d['extra']['c4_mixed_lotd'].get('roofline',{}).get('frac'))"
