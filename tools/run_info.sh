# information pass: ordered launch list of one bench step + isolated kernel timings at batch 1024
TAG=${1:-r03info}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG; O=gpurun_out/$TAG
rm -rf $O/prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-probe > $O/prof.log 2>&1; tail -1 $O/prof.log | cut -c1-300
python tools/step_trace.py $O/prof > $O/step_trace.txt 2>&1; head -3 $O/step_trace.txt
python tools/prof_summary.py $O/prof 60 60 > $O/prof_summary.txt 2>&1
(timeout 600 python tools/kernel_bench.py --batch 1024 --reps 3 --only gemm,dw 2>&1 | tail -80) > $O/kernel_bench.txt; tail -3 $O/kernel_bench.txt
find $O -name "*.csv" -size +20M -delete; du -sh $O
