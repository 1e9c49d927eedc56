cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03f
for round in 1 2; do for pr in 0 1; do
  CVH_MAIN_PRIO=$pr timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prio=$pr', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r03f/prio.txt
done; done
for pr in 0 1; do CVH_ASYNC_DW=0 CVH_MAIN_PRIO=$pr timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('noasync prio=$pr', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r03f/prio.txt; done
