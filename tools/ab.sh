# same-box A/B of library builds (box-to-box variance on the pool exceeds most deltas): bash tools/ab.sh [tag] libA.so libB.so ...
# each library: bench.py twice, interleaved; prints ms per step
TAG=${1:-ab}; shift
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG
for round in 1 2; do
  for lib in "$@"; do
    n=$(basename $lib .so)
    case $n in *base*) export CVH_LN_FORK=0;; *) export CVH_LN_FORK=1;; esac  # builds older than the LayerNorm fork lack its entry point
    CVNETS_HIP_LIB=$GRAFT_REPO_ROOT/ml-cvnets_amd/lib/$lib timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], d['value'])" | tee -a gpurun_out/$TAG/ab.txt
  done
done
