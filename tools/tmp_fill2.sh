cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_fused_ir_gpu.py tests/test_model_gpu.py tests/test_stem_gpu.py tests/test_determinism_gpu.py -q -x 2>&1 | tail -3
rm -rf gpurun_out/tmpprof; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tmpprof -o bench -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-probe > /dev/null 2>&1
grep "at::native" gpurun_out/tmpprof/bench_kernel_stats.csv | cut -d, -f1-3 | cut -c1-150
python tools/prof_summary.py gpurun_out/tmpprof 5 3 | head -3
rm -rf gpurun_out/tmpprof
