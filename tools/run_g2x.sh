cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pb; O=gpurun_out/pb
timeout 1200 python -m pytest tests/test_dwx_gpu.py tests/test_fused_ir_gpu.py tests/test_bench_scale_gpu.py tests/test_determinism_gpu.py tests/test_ir_bwd_gpu.py tests/test_bf16_parity_gpu.py tests/test_model_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee $O/tests.log
bash tools/ab_env.sh pb CVH_IR_PB=0 CVH_IR_PB=1
for v in 0 1; do
rm -rf $O/prof$v; CVH_IR_PB=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$v -o bench -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-probe > $O/prof$v.log 2>&1
python tools/prof_summary.py $O/prof$v 70 0 > $O/prof_summary$v.txt 2>&1
find $O/prof$v -name "*.csv" -size +1M -delete
done
