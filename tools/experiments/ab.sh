# same-box A/B of lib/libcvnets_hip_prev.so (A) vs lib/libcvnets_hip.so (B) on the headline bench; extra env via arguments
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { (timeout 200 env "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline --batch 1024 2>&1 | tail -1) > gpurun_out/b_tmp.log; echo "$@" $(cut -c1-1500 gpurun_out/b_tmp.log | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"total_ms_per_step": [0-9.]*'); }
for i in 1 2; do
run CVNETS_HIP_LIB=$GRAFT_REPO_ROOT/ml-cvnets_amd/lib/libcvnets_hip_prev.so "$@"
run B=1 "$@"
done
