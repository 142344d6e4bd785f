cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_seg_train_gpu.py tests/test_configs_gpu.py::test_config3_64_frames_token_count_and_8_way_chunks tests/test_fullsize_gpu.py::test_encoder_is_independent_per_aligned_frame_chunk tests/test_parallel_gpu.py tests/test_kernels_gpu.py -m gpu -q 2>&1 | tail -3
