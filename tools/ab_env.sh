# same-box A/B of run-time switches: bash tools/ab_env.sh <tag> "VAR=a" "VAR=b" ...   (each setting: bench.py twice, interleaved)
TAG=${1:-abenv}; shift
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG
for round in 1 2; do
  for kv in "$@"; do
    env $kv timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-probe 2>gpurun_out/$TAG/err_$round.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$kv', d['ms_per_step'], d['value'])" | tee -a gpurun_out/$TAG/ab.txt
  done
done
