cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04a
timeout 300 python tools/dwx_debug.py > gpurun_out/r04a/dwx_debug.log 2>&1; tail -40 gpurun_out/r04a/dwx_debug.log
timeout 600 python -m pytest tests/test_dwx_gpu.py -q -m gpu 2>&1 | tail -40 > gpurun_out/r04a/test_dwx.log; tail -15 gpurun_out/r04a/test_dwx.log
bash tools/ab_env.sh r04a CVH_IR_X=0 CVH_IR_X=fwd CVH_IR_X=1
