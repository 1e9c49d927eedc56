cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r05o; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-kernel-probe"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'])"; }
for round in 1 2 3; do
  for v in base wrap oldpb; do
    L=$GRAFT_REPO_ROOT/ml-cvnets_amd/lib/libcvnets_hip_$v.so; [ $v = base ] && L=$GRAFT_REPO_ROOT/ml-cvnets_amd/lib/libcvnets_hip.so
    CVNETS_HIP_LIB=$L $B --batch 1024 --steps 10 --warmup 3 2>/dev/null | line "${v}_b1024" | tee -a $O/ab.txt
  done
done
