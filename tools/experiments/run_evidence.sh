# round evidence: full GPU suite, smoke, default bench line, rocprofv3 kernel stats of the bench command, PMC traffic passes
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG; O=gpurun_out/$TAG
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^  File" | tail -5) > $O/tests_gpu.log; tail -2 $O/tests_gpu.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) > $O/smoke.log; cat $O/smoke.log
(timeout 900 python bench.py 2>&1 | tail -1) > $O/bench.json; cut -c1-400 $O/bench.json
rm -rf $O/prof; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --no-cpu-baseline > $O/prof.log 2>&1; tail -1 $O/prof.log | cut -c1-200
python tools/prof_summary.py $O/prof 40 30 > $O/prof_summary.txt 2>&1; python tools/timeline.py $O/prof > $O/timeline.txt 2>&1; head -4 $O/timeline.txt
for c in FETCH_SIZE WRITE_SIZE; do rm -rf $O/pmc_$c; timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python bench.py --steps 2 --warmup 1 --no-kernel-probe --no-cpu-baseline > $O/pmc_$c.log 2>&1; done
python tools/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_summary.txt 2>&1; head -3 $O/pmc_summary.txt
python tools/make_pmc_json.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/prof 1024 $TAG > $O/pmc_traffic.json 2>$O/pmc_json.err; head -12 $O/pmc_traffic.json
find $O -name "*counter_collection.csv" -size +30M -delete; find $O -name "*kernel_trace.csv" -size +30M -delete; du -sh $O
