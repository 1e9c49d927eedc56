cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04e; O=gpurun_out/r04e
for sh in 2 1; do for d in 0 1 2 4 8 16 3 12 6 10 28 30; do echo -n "shape $sh dbg $d: "; timeout 120 python tools/bench_dwx.py --only new --shape $sh --dbg $d --reps 3 2>&1 | grep -v amdgpu.ids; done; done | tee $O/knobs.log
