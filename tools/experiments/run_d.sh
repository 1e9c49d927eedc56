cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03d
timeout 1500 python -m pytest tests/test_advice_gpu.py tests/test_kernels_gpu.py tests/test_dropin_gpu.py tests/test_next_rows_gpu.py tests/test_bench_scale_gpu.py -q -m gpu -x > gpurun_out/r03d/t.log 2>&1; tail -3 gpurun_out/r03d/t.log
bash tools/ab.sh r03d libcvnets_hip_base.so libcvnets_hip.so
