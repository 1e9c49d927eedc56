cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04l; O=gpurun_out/r04l
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/tests_gpu.log; tail -6 $O/tests_gpu.log
rm -rf $O/prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-probe > $O/prof.log 2>&1; tail -1 $O/prof.log | cut -c1-200
python tools/prof_summary.py $O/prof 70 30 > $O/prof_summary.txt 2>&1; python tools/step_trace.py $O/prof > $O/step_trace.txt 2>&1
head -45 $O/prof_summary.txt
find $O -name "*kernel_trace.csv" -size +8M -delete
