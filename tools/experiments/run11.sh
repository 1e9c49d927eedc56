cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_fused_ir_gpu.py -x -q -m gpu 2>&1 | grep -v "^  File" | tail -3) > gpurun_out/t_sk.log; cat gpurun_out/t_sk.log
run() { (timeout 600 env "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline --batch 1024 2>&1 | tail -1) > gpurun_out/b_tmp.log; echo "$@" $(cut -c1-200 gpurun_out/b_tmp.log | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*'); }
run CVH_FOLD_BIAS=1
run CVH_ASYNC_DW=0
run CVH_FOLD_BIAS=1
