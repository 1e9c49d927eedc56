cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r03pmca; mkdir -p $O
rm -rf $O/p1; timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/p1 -o m -- python tools/experiments/pmc_attn.py > $O/p1.log 2>&1
rm -rf $O/p2; timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/p2 -o m -- python tools/experiments/pmc_attn.py > $O/p2.log 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/r03pmca/p1", "gpurun_out/r03pmca/p2"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    agg = collections.OrderedDict()
    for r in rows:
        agg.setdefault((r["Dispatch_Id"], r["Kernel_Name"][:40]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    seen = collections.Counter()
    for (did, name), c in agg.items():
        if "attn" not in name: continue
        seen[name] += 1
        if seen[name] == 2: print(name, {k: int(v) for k, v in c.items()})
PY
