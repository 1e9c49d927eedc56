cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04c; O=gpurun_out/r04c
timeout 600 python -m pytest tests/test_dwx_gpu.py -q -m gpu 2>&1 | tail -5 > $O/test_dwx.log; tail -5 $O/test_dwx.log
timeout 300 python tools/bench_dwx.py > $O/bench_dwx.log 2>&1; cat $O/bench_dwx.log
rm -rf $O/pmc1; timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc1 -o pmc -- python tools/bench_dwx.py --reps 1 --only new > $O/pmc1.log 2>&1
rm -rf $O/pmc2; timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $O/pmc2 -o pmc -- python tools/bench_dwx.py --reps 1 --only new > $O/pmc2.log 2>&1
python - <<'PY' > $O/pmc_dwx.txt 2>&1
import csv, glob, collections
for d in ("gpurun_out/r04c/pmc1", "gpurun_out/r04c/pmc2"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f:
        print(d, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:40] + " grid" + r.get("Grid_Size", "")
        if "dwx" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in agg:
        print(k)
        for c, v in sorted(agg[k].items()):
            print(f"    {c:28s} {v / n[(k, c)]:.4g}")
PY
cat $O/pmc_dwx.txt | head -150
bash tools/ab_env.sh r04c CVH_IR_X=0 CVH_IR_X=1
find $O -name "*.csv" -size +4M -delete; du -sh $O
