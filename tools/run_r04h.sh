cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04h; O=gpurun_out/r04h
rocprofv3 -L 2>/dev/null | grep -oE "^\s*(Name|name)\s*:\s*\S+|^\S+ *$" | head -5
rocprofv3 -L > $O/counters.txt 2>&1; grep -c . $O/counters.txt; grep -oE "\b(TA|TCP|TCC|SQ|TD|GRBM)_[A-Za-z0-9_]+" $O/counters.txt | sort -u > $O/counter_names.txt; wc -l $O/counter_names.txt
run() { tag=$1; shift; rm -rf $O/$tag; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$tag -o pmc -- python tools/bench_dwx.py --reps 1 --shape 2 > $O/$tag.log 2>&1; }
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run p2 SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES
run p3 TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum
run p4 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum
run p5 GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_READ_sum TCC_WRITE_sum
python - <<'PY' > $O/pmc.txt 2>&1
import csv, glob, collections
for d in ("p1","p2","p3","p4","p5"):
    f = glob.glob("gpurun_out/r04h/%s/**/*counter_collection.csv" % d, recursive=True)
    if not f:
        print(d, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        k = k[k.find("dw"):k.find("dw")+22] if "dw" in k else (k[k.find("gemm_stream"):][:22] if "gemm_stream" in k else None)
        if k is None: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in sorted(agg):
        print(d, k, "  ".join(f"{c}={v / n[(k, c)]:.4g}" for c, v in sorted(agg[k].items())))
PY
cat $O/pmc.txt
grep -iE "error|invalid|not" $O/p3.log $O/p4.log $O/p5.log | head
find $O -name "*.csv" -size +2M -delete
