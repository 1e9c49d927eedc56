cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r05n; mkdir -p $O
for v in base wrap oldpb; do
  L=$GRAFT_REPO_ROOT/ml-cvnets_amd/lib/libcvnets_hip_$v.so; [ $v = base ] && L=$GRAFT_REPO_ROOT/ml-cvnets_amd/lib/libcvnets_hip.so
  rm -rf $O/prof_$v; CVNETS_HIP_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-probe > $O/prof_$v.log 2>&1
  python tools/prof_summary.py $O/prof_$v 70 3 > $O/prof_summary_$v.txt 2>&1
  echo "== $v: $(head -1 $O/prof_summary_$v.txt)"; grep -E "ir_pb_kernel|gemm_stream_kernel|ir_exp_bwd_kernel|ir_red_fwd_kernel" $O/prof_summary_$v.txt | head -16
done
find $O -name "*kernel_trace.csv" -size +8M -delete
