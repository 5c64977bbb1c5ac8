python -m pytest tests/test_lotd_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error" | tail -3
bash tools/gpu_variants.sh r02p_align "NR3D_PAIR_ALIGN=0,1" "NR3D_PAIR_FIXED=0,1" "NR3D_PAIR_EPB_LOG2=12,13"
