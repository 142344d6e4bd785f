cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/lab
timeout 1200 python3 -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_stages_gpu.py -m gpu -x -q -s 2>&1 | grep -E "TOWER_STREAM|passed|failed|Error|error" | tail -8
