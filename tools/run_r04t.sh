cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04t; O=gpurun_out/r04t
for sh in 2 0; do for d in 0 2 4 8 16 6 12 24 28 0; do echo -n "shape $sh dbg $d: "; timeout 120 python tools/bench_dwx.py --only new --shape $sh --dbg $d --reps 4 2>&1 | grep -v amdgpu.ids | cut -c1-60; done; done | tee $O/knobs.log
