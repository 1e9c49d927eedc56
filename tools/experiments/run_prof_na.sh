TAG=${1:-r03pna}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG; O=gpurun_out/$TAG
export CVH_ASYNC_DW=0
rm -rf $O/prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-probe > $O/prof.log 2>&1; tail -1 $O/prof.log | cut -c1-120
python tools/step_trace.py $O/prof > $O/step_trace.txt 2>&1; python tools/prof_summary.py $O/prof 80 10 > $O/prof_summary.txt 2>&1
find $O -name "*.csv" -size +20M -delete
