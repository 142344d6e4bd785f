cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/lab
python3 tools/lab/attn_vit_membound.py 2>&1 | grep -v amdgpu.ids
bash tools/lab/ktrace.sh attn tools/lab/attn_vit_membound.py
