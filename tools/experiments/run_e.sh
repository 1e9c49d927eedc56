cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r03e; mkdir -p $O
timeout 300 python tools/experiments/pmc_micro2.py time 2>&1 | grep "K=" | tee $O/align.txt
rm -rf $O/p1; timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/p1 -o m -- python tools/experiments/pmc_micro2.py > $O/p1.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r03e/p1/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = collections.OrderedDict()
for r in rows:
    agg.setdefault((r["Dispatch_Id"], r["Kernel_Name"][:60]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for (did, name), c in agg.items():
    if "gemm_stream" in name: print(name, {k: int(v) for k, v in c.items()})
PY
