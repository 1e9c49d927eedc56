cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu 2>&1 | tail -2) > gpurun_out/t_k.log; cat gpurun_out/t_k.log
rm -rf gpurun_out/prof_v; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v -o vit -- python tools/bench_models.py --models vit_base --batch vit_base=512 --steps 3 --warmup 1 > gpurun_out/prof_v.log 2>&1; grep "^{" gpurun_out/prof_v.log | cut -c1-200
python tools/timeline.py gpurun_out/prof_v > gpurun_out/timeline_v.txt 2>&1; head -30 gpurun_out/timeline_v.txt
find gpurun_out/prof_v -name "*kernel_trace.csv" -size +30M -delete
