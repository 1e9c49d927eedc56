# round 5, call 3: the full GPU suite three times in the default order (dropout step tests in the alphabetical middle, no cache-drop fixture,
# references without MIOpen), the guard sweep in "begin" mode (accesses BEFORE the first byte of an operand), measured eval-mode errors
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c; mkdir -p $O
for i in 1 2 3; do
  timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/full_$i.log 2>&1; echo "run $i: $(grep -E ' passed| failed|Memory access|Aborted' $O/full_$i.log | tail -2 | tr '\n' ' ')"
done
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) > $O/smoke.log; cat $O/smoke.log
timeout 300 python -m pytest tests/test_model_gpu.py tests/test_bench_scale_gpu.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep "^\[" > $O/eval_errors.txt; cat $O/eval_errors.txt | cut -c1-220
python tools/guard_run.py $O/guard_begin --budget 420 --mode begin > $O/guard_begin_summary.txt 2>&1; cat $O/guard_begin_summary.txt | cut -c1-600
