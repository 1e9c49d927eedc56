cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04k; O=gpurun_out/r04k
timeout 600 python -m pytest tests/test_dwx_gpu.py -q -m gpu 2>&1 | tail -4 > $O/test_dwx.log; tail -4 $O/test_dwx.log
echo "--- occ (3,3)"; timeout 300 python tools/bench_dwx.py --only new 2>&1 | grep -v amdgpu.ids | tee $O/bench_dwx_A.log
echo "--- occ (4,2)"; CVNETS_HIP_LIB=$GRAFT_REPO_ROOT/ml-cvnets_amd/lib/libcvnets_hip_occB.so timeout 300 python tools/bench_dwx.py --only new 2>&1 | grep -v amdgpu.ids | tee $O/bench_dwx_B.log
echo "--- occ (3,3)"; timeout 300 python tools/bench_dwx.py --only new 2>&1 | grep -v amdgpu.ids | tee -a $O/bench_dwx_A.log
timeout 900 python -m pytest tests/test_bf16_parity_gpu.py -q -m gpu -s -k "rounding_points" 2>&1 | grep -E "bf16 points|passed|failed|Error|assert" | tee $O/points.log
timeout 300 python tools/bench_engine.py --graph-compare 2>&1 | grep -v amdgpu.ids | tee $O/engine.log
