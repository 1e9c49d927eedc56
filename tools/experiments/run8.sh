cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo skip tests
run() { (timeout 600 env "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline --batch 1024 2>&1 | tail -1) > gpurun_out/b_tmp.log; echo "$@" $(cut -c1-200 gpurun_out/b_tmp.log | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*'); }
run CVH_TUNE=9=1 CVH_FOLD_BIAS=0
run CVH_TUNE=9=0 CVH_FOLD_BIAS=0
run CVH_TUNE=9=1 CVH_FOLD_BIAS=1
run CVH_TUNE=9=0 CVH_FOLD_BIAS=1
run CVH_TUNE=10=512 CVH_FOLD_BIAS=1
run CVH_TUNE=10=1280 CVH_FOLD_BIAS=1
run CVH_TUNE=9=1 CVH_FOLD_BIAS=0
