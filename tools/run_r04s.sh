cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04s; O=gpurun_out/r04s
timeout 900 python -m pytest tests/test_dwx_gpu.py tests/test_fused_ir_gpu.py tests/test_model_gpu.py tests/test_determinism_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee $O/tests.log
for i in 1 2; do timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; done | tee $O/bench.log
rm -rf $O/prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-probe > $O/prof.log 2>&1
python tools/prof_summary.py $O/prof 12 5 2>&1 | head -3
python tools/step_trace.py $O/prof 2>/dev/null | sed -n 12,24p
find $O -name "*kernel_trace.csv" -size +8M -delete
