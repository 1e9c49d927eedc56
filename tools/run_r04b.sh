cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04b; O=gpurun_out/r04b
timeout 300 python tools/dwx_debug2.py 16 32 1 40 > $O/debug2_a.log 2>&1; tail -30 $O/debug2_a.log
timeout 300 python tools/dwx_debug2.py 64 64 1 32 > $O/debug2_b.log 2>&1; tail -22 $O/debug2_b.log
rm -rf $O/prof; CVH_IR_X=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-probe > $O/prof.log 2>&1; tail -1 $O/prof.log | cut -c1-200
python tools/prof_summary.py $O/prof 70 30 > $O/prof_summary.txt 2>&1; python tools/step_trace.py $O/prof > $O/step_trace.txt 2>&1
grep -E "dwx|dwf|gemm_stream_kernel<2, 2, 1>|skinny|colreduce" $O/prof_summary.txt | head -40
find $O -name "*kernel_trace.csv" -size +8M -delete; du -sh $O
