cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04b
python -m pytest tests/test_mlp_gpu.py -m gpu -x -q -k "half" 2>&1 | tail -15 > gpurun_out/r04b/pytest_half.log; cat gpurun_out/r04b/pytest_half.log
python - <<'PY' 2>&1 | tail -5
import sys, json, torch
sys.path.insert(0, '.')
import bench
print(json.dumps(bench.mlp_decoder_rate(torch.device('cuda:0'))))
PY
