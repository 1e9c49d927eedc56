cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03d
timeout 900 python -m pytest tests/test_gemm_stream_gpu.py -q -m gpu -x > gpurun_out/r03d/t.log 2>&1; tail -3 gpurun_out/r03d/t.log
for lib in libcvnets_hip_sb.so libcvnets_hip.so; do echo $lib; CVNETS_HIP_LIB=$GRAFT_REPO_ROOT/ml-cvnets_amd/lib/$lib timeout 300 python tools/experiments/pmc_micro2.py time 2>&1 | grep "K="; done | tee gpurun_out/r03d/align.txt
bash tools/ab.sh r03d libcvnets_hip_sb.so libcvnets_hip.so
