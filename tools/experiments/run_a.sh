cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03a
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r03a/full.log 2>&1; tail -3 gpurun_out/r03a/full.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-probe > gpurun_out/r03a/bench.log 2>&1; tail -1 gpurun_out/r03a/bench.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r03a
rm -rf $O/prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-probe > $O/prof.log 2>&1
python tools/step_trace.py $O/prof > $O/step_trace.txt 2>&1; python tools/prof_summary.py $O/prof 60 10 > $O/prof_summary.txt 2>&1; head -3 $O/prof_summary.txt
find $O -name "*.csv" -size +20M -delete
