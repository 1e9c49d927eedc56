cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04n; O=gpurun_out/r04n
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "linear_fwd_bwd or big_gemm" 2>&1 | tail -8 | tee $O/tests.log
timeout 600 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tee $O/bench_gemm.log
timeout 900 python tools/bench_models.py --models vit_base,clip --batch vit_base=512,clip=256 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-400 | tee $O/models.log
rm -rf $O/prof128; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof128 -o bench -- python bench.py --batch 128 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-probe > $O/prof128.log 2>&1
python tools/prof_summary.py $O/prof128 60 10 > $O/prof128_summary.txt 2>&1; head -50 $O/prof128_summary.txt
find $O -name "*kernel_trace.csv" -size +8M -delete
