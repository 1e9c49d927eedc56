cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03b
timeout 900 python -m pytest tests/test_fused_ir_gpu.py tests/test_bench_scale_gpu.py tests/test_model_gpu.py -q -m gpu -x > gpurun_out/r03b/t.log 2>&1; tail -3 gpurun_out/r03b/t.log
timeout 300 python tools/experiments/bench_dwf.py 1024 > gpurun_out/r03b/bench_dwf.txt 2>&1
grep "fwd\|full" gpurun_out/r03b/bench_dwf.txt
bash tools/ab.sh r03b libcvnets_hip_base.so libcvnets_hip.so
