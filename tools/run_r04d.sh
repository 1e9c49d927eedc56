cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04d; O=gpurun_out/r04d
timeout 600 python -m pytest tests/test_dwx_gpu.py -q -m gpu 2>&1 | tail -8 > $O/test_dwx.log; tail -8 $O/test_dwx.log
echo "--- occ A (4/3)"; timeout 300 python tools/bench_dwx.py --only new 2>&1 | grep -v amdgpu.ids | tee $O/bench_dwx_A.log
echo "--- occ B (3/2)"; CVNETS_HIP_LIB=$GRAFT_REPO_ROOT/ml-cvnets_amd/lib/libcvnets_hip_occB.so timeout 300 python tools/bench_dwx.py --only new 2>&1 | grep -v amdgpu.ids | tee $O/bench_dwx_B.log
bash tools/ab_env.sh r04d CVH_IR_X=0 CVH_IR_X=1
