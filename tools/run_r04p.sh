cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04p; O=gpurun_out/r04p
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "linear_fwd_bwd or big_gemm" 2>&1 | tail -3 | tee $O/tests.log
timeout 600 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tee $O/bench_gemm.log
