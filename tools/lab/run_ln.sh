cd $GRAFT_REPO_ROOT
python3 tools/lab/connector_ops_time.py 2>&1 | grep -E "ln\+"
echo NO_LDS; UFV_LN_PACKED_NO_LDS=1 python3 tools/lab/connector_ops_time.py 2>&1 | grep -E "ln\+silu"
python3 -m pytest tests/test_kernels_gpu.py -m gpu -q -k "layernorm or ln_add" 2>&1 | tail -1
