cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/lab
bash tools/lab/ktrace.sh se tools/lab/connector_ops_time.py
