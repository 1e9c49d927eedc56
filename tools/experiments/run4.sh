cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for b in 128 256 512 1024; do (timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --batch $b 2>&1 | tail -1) > gpurun_out/b_batch$b.log; echo batch $b $(cut -c1-200 gpurun_out/b_batch$b.log | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*'); done
python -c "import torch; print(torch.cuda.max_memory_allocated())"
