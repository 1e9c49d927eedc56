cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04q; O=gpurun_out/r04q
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "linear_fwd_bwd or big_gemm" 2>&1 | tail -3 | tee $O/tests.log
run() { tag=$1; shift; rm -rf $O/$tag; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$tag -o pmc -- python tools/bench_dwx.py --reps 1 --only new > $O/$tag.log 2>&1; }
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS
run p2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS SQ_WAVES
run p3 TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE SQ_INSTS_MFMA
python - <<'PY' | tee $O/pmc.txt
import csv, glob, collections
for d in ("p1","p2","p3"):
    f = glob.glob("gpurun_out/r04q/%s/**/*counter_collection.csv" % d, recursive=True)
    if not f: print(d, "none"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "dwx" not in k: continue
        k = k[k.find("dwx"):k.find("dwx")+22]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in sorted(agg):
        print(d, k, "  ".join(f"{c}={v / n[(k, c)]:.4g}" for c, v in sorted(agg[k].items())))
PY
find $O -name "*.csv" -size +2M -delete
