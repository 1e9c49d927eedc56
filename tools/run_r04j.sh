cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04j; O=gpurun_out/r04j
timeout 600 python -m pytest tests/test_dwx_gpu.py -q -m gpu 2>&1 | tail -12 > $O/test_dwx.log; tail -12 $O/test_dwx.log
timeout 300 python tools/bench_dwx.py 2>&1 | grep -v amdgpu.ids | tee $O/bench_dwx.log
bash tools/ab_env.sh r04j CVH_IR_X=0 CVH_IR_X=1
timeout 900 python -m pytest tests/test_bf16_parity_gpu.py -q -m gpu -s -k "rounding_points" 2>&1 | grep -E "bf16 points|passed|failed|Error|assert" | tee $O/points.log
