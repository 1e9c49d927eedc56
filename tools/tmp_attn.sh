cd $GRAFT_REPO_ROOT
for round in 1 2; do for t in "25=1" "25=0" "25=2" "25=3"; do
CVH_TUNE=$t timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tune $t', d['ms_per_step'], d['value'])"
done; done
rocm-smi --showclocks 2>/dev/null | head -20
