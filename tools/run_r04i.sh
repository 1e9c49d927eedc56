cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04i; O=gpurun_out/r04i
timeout 600 python -m pytest tests/test_dwx_gpu.py -q -m gpu 2>&1 | tail -3 > $O/test_dwx.log; tail -3 $O/test_dwx.log
for n in 0 1 0 1; do echo "--- narrow $n"; timeout 300 python tools/bench_dwx.py --only new --narrow $n 2>&1 | grep -v amdgpu.ids; done | tee $O/wide.log
