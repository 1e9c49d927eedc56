cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 150 python -m pytest tests/test_dw_skinny_gpu.py -x -q -m gpu 2>&1 | grep -v "^  File" | tail -3) > gpurun_out/t_sk.log; cat gpurun_out/t_sk.log
grep -q "passed" gpurun_out/t_sk.log || exit 0
grep -q "failed\|error" gpurun_out/t_sk.log && exit 0
(timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_fused_ir_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | grep -v "^  File" | tail -3) > gpurun_out/t_sk2.log; cat gpurun_out/t_sk2.log
grep -q "failed\|error" gpurun_out/t_sk2.log && exit 0
run() { (timeout 200 env "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline --batch 1024 2>&1 | tail -1) > gpurun_out/b_tmp.log; echo "$@" $(cut -c1-1500 gpurun_out/b_tmp.log | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"total_ms_per_step": [0-9.]*'); }
run CVNETS_HIP_LIB=$GRAFT_REPO_ROOT/ml-cvnets_amd/lib/libcvnets_hip_prev.so
run A=1
run CVNETS_HIP_LIB=$GRAFT_REPO_ROOT/ml-cvnets_amd/lib/libcvnets_hip_prev.so
run A=1
