cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03t
timeout 600 python -m pytest tests/test_rccl_gpu.py -q -m gpu > gpurun_out/r03t/rccl.log 2>&1
tail -5 gpurun_out/r03t/rccl.log
