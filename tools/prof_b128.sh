# kernel table of the 128-image step (the 8-GPU strong-scaling shard) on one GPU:   bash tools/prof_b128.sh <tag>
TAG=${1:-b128}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG; O=gpurun_out/$TAG
rm -rf $O/prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --batch 128 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-probe > $O/prof.log 2>&1; tail -1 $O/prof.log | cut -c1-200
python tools/prof_summary.py $O/prof 70 60 > $O/prof_summary.txt 2>&1; python tools/timeline.py $O/prof > $O/timeline.txt 2>&1; head -3 $O/timeline.txt
python tools/step_trace.py $O/prof > $O/step_trace.txt 2>&1
find $O -name "*kernel_trace.csv" -size +8M -delete
(timeout 300 python bench.py --batch 128 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-probe 2>/dev/null | tail -1) > $O/bench_b128.json; cut -c1-200 $O/bench_b128.json
