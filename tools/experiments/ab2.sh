cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { (timeout 200 env "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-probe --batch 1024 2>&1 | tail -1) > gpurun_out/b_tmp.log; echo "$@" $(cut -c1-1500 gpurun_out/b_tmp.log | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*'); }
for i in 1 2; do
run CVH_TUNE=11=1
run CVH_TUNE=12=64
run CVH_TUNE=12=128
run CVH_TUNE=12=160
done
