cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^  File" | tail -4) > gpurun_out/t_all.log; cat gpurun_out/t_all.log
for b in 1024 128; do (timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --batch $b 2>&1 | tail -1) > gpurun_out/b_batch$b.log; echo batch $b $(cut -c1-200 gpurun_out/b_batch$b.log | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*'); done
bash tools/experiments/run_pmc.sh 1024 | head -24
