cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03d
timeout 900 python -m pytest tests/test_gemm_stream_gpu.py tests/test_bench_scale_gpu.py tests/test_fused_ir_gpu.py -q -m gpu -x > gpurun_out/r03d/t.log 2>&1; tail -5 gpurun_out/r03d/t.log
timeout 300 python tools/experiments/bench_fx.py 2>&1 | grep "plain+stats  \|e_mode1+stats  " > gpurun_out/r03d/fx.txt; cat gpurun_out/r03d/fx.txt
bash tools/ab.sh r03d libcvnets_hip_base.so libcvnets_hip.so
