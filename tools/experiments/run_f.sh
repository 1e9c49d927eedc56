cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03f
for round in 1 2; do for fork in 0 1; do
  CVH_LN_FORK=$fork timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fork=$fork', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r03f/ab.txt
done; done
for lib in libcvnets_hip_base.so libcvnets_hip.so; do echo $lib; CVH_LN_FORK=0 CVNETS_HIP_LIB=$GRAFT_REPO_ROOT/ml-cvnets_amd/lib/$lib timeout 300 python tools/kernel_bench.py --batch 1024 --reps 5 --only attn 2>&1 | grep attn; done | tee gpurun_out/r03f/attn.txt
