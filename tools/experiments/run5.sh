cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B=${1:-1024}
rm -rf gpurun_out/prof_b; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b -o big -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --batch $B > gpurun_out/prof_b.log 2>&1; tail -1 gpurun_out/prof_b.log | cut -c1-200; python tools/prof_summary.py gpurun_out/prof_b 40 30 > gpurun_out/prof_b_summary.txt 2>&1; cat gpurun_out/prof_b_summary.txt
