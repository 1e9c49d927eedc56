cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04g; O=gpurun_out/r04g
timeout 600 python -m pytest tests/test_dwx_gpu.py -q -m gpu 2>&1 | tail -3 > $O/test_dwx.log; tail -3 $O/test_dwx.log
for sg in 0 1 2 3 5 8; do echo "--- stagger $sg"; timeout 300 python tools/bench_dwx.py --only new --stagger $sg 2>&1 | grep -v amdgpu.ids; done | tee $O/stagger.log
