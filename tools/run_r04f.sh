cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04f; O=gpurun_out/r04f
timeout 600 python -m pytest tests/test_dwx_gpu.py -q -m gpu 2>&1 | tail -12 > $O/test_dwx.log; tail -12 $O/test_dwx.log
timeout 300 python tools/bench_dwx.py --only new 2>&1 | grep -v amdgpu.ids | tee $O/bench_dwx.log
for d in 2 4 8 16 28; do echo -n "shape 2 dbg $d: "; timeout 120 python tools/bench_dwx.py --only new --shape 2 --dbg $d --reps 3 2>&1 | grep -v amdgpu.ids; done | tee $O/knobs.log
bash tools/ab_env.sh r04f CVH_IR_X=0 CVH_IR_X=1
