cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04o; O=gpurun_out/r04o
cat > /tmp/g1.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "ml-cvnets_amd"))
from cvnets_amd import _lib, ops
ops.set_compute_dtype(torch.bfloat16)
M, K, N = 100864, 768, 3072
x = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda") * K ** -0.5; b = torch.randn(N, device="cuda")
with torch.no_grad():
    for knob in (1, 3):
        _lib.call("cvh_set_tuning", 5, knob)
        for _ in range(3): ops.linear(x, w, b)
    torch.nn.functional.linear(x, w.bfloat16(), b.bfloat16()); torch.nn.functional.linear(x, w.bfloat16(), b.bfloat16())
torch.cuda.synchronize()
PY
run() { tag=$1; shift; rm -rf $O/$tag; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$tag -o pmc -- python /tmp/g1.py > $O/$tag.log 2>&1; }
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run p2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVES
run p3 TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
run p4 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
python - <<'PY' | tee $O/pmc.txt
import csv, glob, collections
for d in ("p1","p2","p3","p4"):
    f = glob.glob("gpurun_out/r04o/%s/**/*counter_collection.csv" % d, recursive=True)
    if not f: print(d, "none"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        k = "nt256" if "nt256" in k else ("nt128" if "nt128" in k else (k[:40] if ("Cijk" in k or "gemm" in k.lower()) else None))
        if k is None: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in sorted(agg):
        print(d, k, "  ".join(f"{c}={v / n[(k, c)]:.4g}" for c, v in sorted(agg[k].items())))
PY
grep -h "Kernel_Name" -m1 $O/p1/*/*kernel_trace.csv 2>/dev/null | head -2; python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r04o/p1/**/*kernel_trace.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    n = r["Kernel_Name"]
    if "nt256" in n or "nt128" in n or "Cijk" in n:
        print(n[:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us", "vgpr", r.get("VGPR_Count"), "accum", r.get("Accum_VGPR_Count"), "lds", r.get("LDS_Block_Size"), "grid", r.get("Grid_Size"), "wg", r.get("Workgroup_Size"))
PY
find $O -name "*.csv" -size +2M -delete
