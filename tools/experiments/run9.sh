cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_fused_ir_gpu.py tests/test_dw_skinny_gpu.py -x -q -m gpu 2>&1 | grep -v "^  File" | tail -3) > gpurun_out/t_sk.log; cat gpurun_out/t_sk.log
run() { (timeout 600 env "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline --batch 1024 2>&1 | tail -1) > gpurun_out/b_tmp.log; echo "$@" $(cut -c1-200 gpurun_out/b_tmp.log | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*'); }
run CVH_FOLD_BIAS=1
run CVH_FOLD_BIAS=1
bash tools/experiments/run5.sh 1024 > gpurun_out/run5.log 2>&1
python tools/timeline.py gpurun_out/prof_b > gpurun_out/timeline.txt; head -30 gpurun_out/timeline.txt
