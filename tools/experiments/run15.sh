cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 300 python -m pytest tests/test_widened_rows_gpu.py -x -q -m gpu -k "attention or attn" 2>&1 | grep -v "^  File" | tail -25) > gpurun_out/t_ad.log; cat gpurun_out/t_ad.log
(timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attn or attention or mha" 2>&1 | tail -3)
