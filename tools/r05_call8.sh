cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r05h; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-kernel-probe"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'])"; }
for round in 1 2; do
  for kv in "0=0" "6=256" "6=512" "6=2048" "3=256" "3=1024" "2=256" "2=1024" "10=256"; do
    CVH_TUNE="$kv" $B --batch 128 --steps 40 --warmup 3 2>/dev/null | line "tune_${kv}_b128" | tee -a $O/ab.txt
  done
  for kv in "0=0" "6=2048" "3=1024" "2=1024"; do
    CVH_TUNE="$kv" $B --batch 1024 --steps 10 --warmup 3 2>/dev/null | line "tune_${kv}_b1024" | tee -a $O/ab.txt
  done
done
