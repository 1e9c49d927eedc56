# round 5, call 4: same-box A/B experiments (each setting measured twice, interleaved) + the measurements still missing
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r05d; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-kernel-probe"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'])"; }
for round in 1 2; do
  for b in 1024 128; do
    st=$([ $b = 1024 ] && echo 10 || echo 40)
    $B --batch $b --steps $st --warmup 3 2>/dev/null | line "base_b$b" | tee -a $O/ab.txt
    CVNETS_HIP_LIB=$GRAFT_REPO_ROOT/ml-cvnets_amd/lib/libcvnets_hip_v1.so $B --batch $b --steps $st --warmup 3 2>/dev/null | line "v1_dwxfwd64_3wg_b$b" | tee -a $O/ab.txt
    CVH_ASYNC_DW=1 $B --batch $b --steps $st --warmup 3 2>/dev/null | line "async_dw_b$b" | tee -a $O/ab.txt
  done
  for f in 128 256 512; do CVH_TUNE="18=$f" $B --batch 128 --steps 40 --warmup 3 2>/dev/null | line "gemm_fill${f}_b128" | tee -a $O/ab.txt; done
done
# measured eval-mode errors (tests print them)
timeout 300 python -m pytest tests/test_model_gpu.py tests/test_bench_scale_gpu.py tests/test_bf16_parity_gpu.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -o "\[.*" | grep -v "grad::" > $O/parity_lines.txt; grep -c . $O/parity_lines.txt
# engine-driven loop: torch AdamW vs the launcher's fused AdamW, next to the replayed step
(timeout 600 python tools/bench_engine.py --batch 128,1024 --graph-compare 2>/dev/null | grep "^{") > $O/bench_engine.jsonl; cut -c1-330 $O/bench_engine.jsonl
# new tests of this call's tree
timeout 300 python -m pytest tests/test_launch_gpu.py tests/test_rccl_gpu.py tests/test_fused_ir_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
# the 128-image step's kernel table
rm -rf $O/prof128; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof128 -o bench -- python bench.py --batch 128 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-probe > $O/prof128.log 2>&1
python tools/prof_summary.py $O/prof128 70 10 > $O/b128_prof_summary.txt 2>&1; head -40 $O/b128_prof_summary.txt | cut -c1-150
find $O -name "*kernel_trace.csv" -size +8M -delete; du -sh $O
