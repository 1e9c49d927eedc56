cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03c
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_vit_gpu.py tests/test_clip_gpu.py tests/test_model_gpu.py -q -m gpu -x > gpurun_out/r03c/t.log 2>&1; tail -3 gpurun_out/r03c/t.log
for lib in libcvnets_hip_base.so libcvnets_hip.so; do echo $lib; CVNETS_HIP_LIB=$GRAFT_REPO_ROOT/ml-cvnets_amd/lib/$lib timeout 300 python tools/kernel_bench.py --batch 1024 --reps 5 --only attn 2>&1 | grep attn; done | tee gpurun_out/r03c/attn.txt
