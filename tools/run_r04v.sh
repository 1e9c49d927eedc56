cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04v; O=gpurun_out/r04v
timeout 600 python -m pytest tests/test_dwx_gpu.py tests/test_bench_scale_gpu.py tests/test_determinism_gpu.py -q -m gpu 2>&1 | tail -3 | tee $O/test_dwx.log
for i in 1 2; do timeout 300 python tools/bench_dwx.py --only new 2>&1 | grep -v amdgpu.ids; done | tee $O/bench_dwx.log
bash tools/ab_env.sh r04v CVH_IR_X=0 CVH_IR_X=1
