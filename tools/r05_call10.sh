cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r05j; mkdir -p $O
rm -rf $O/prof_v2; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_v2 -o bench -- python tools/bench_models.py --models mobilevitv2 --batch mobilevitv2=128 --steps 5 --warmup 2 > $O/prof_v2.log 2>&1
python tools/prof_summary.py $O/prof_v2 45 5 > $O/v2_prof_summary.txt 2>&1; head -50 $O/v2_prof_summary.txt | cut -c1-160
find $O -name "*kernel_trace.csv" -size +8M -delete
