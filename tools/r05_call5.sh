cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r05e; mkdir -p $O
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -6) > $O/smoke.log; cat $O/smoke.log
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_bench_scale_gpu.py tests/test_launch_gpu.py tests/test_rccl_gpu.py tests/test_fused_ir_gpu.py tests/test_guard_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4 > $O/tests_subset.log; cat $O/tests_subset.log
B="python bench.py --no-cpu-baseline --no-kernel-probe"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'])"; }
for round in 1 2; do
  $B --batch 1024 --steps 10 --warmup 3 2>/dev/null | line "new_b1024" | tee -a $O/ab.txt
  $B --batch 128 --steps 40 --warmup 3 2>/dev/null | line "new_b128" | tee -a $O/ab.txt
  CVH_TUNE="18=0" $B --batch 128 --steps 40 --warmup 3 2>/dev/null | line "nofill_b128" | tee -a $O/ab.txt
done
