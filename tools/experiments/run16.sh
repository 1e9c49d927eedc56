cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rocprofv3 -L 2>/dev/null | grep "Counter_Name" | grep -E "VALUBusy|MemUnitBusy|MfmaUtil|VALUUtil|SALUBusy|LdsUtil|LDSBankConf|OccupancyPercent|MemUnitStalled|WriteUnitStalled|SQ_INSTS_VALU\b|SQ_ACTIVE_INST_VALU\b|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_WAIT_INST_ANY|SQ_INST_CYCLES_VMEM|SQ_INSTS_LDS|SQ_LDS_BANK_CONFLICT|MeanOccupancy" | head -30
for c in "VALUBusy MemUnitBusy" "MfmaUtil LdsUtil" "MemUnitStalled WriteUnitStalled" "MeanOccupancyPerCU"; do
  n=$(echo $c | tr ' ' '_'); rm -rf gpurun_out/pmcfx_$n
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmcfx_$n -o p -- python tools/experiments/bench_fx.py > gpurun_out/pmcfx_$n.log 2>&1
  python - "$n" <<'PY'
import csv,glob,sys,collections
n=sys.argv[1]
fs=glob.glob(f'gpurun_out/pmcfx_{n}/*counter_collection.csv')
if not fs: print(n,'no output'); sys.exit()
d=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    k=r['Kernel_Name']
    if 'conv_gemm' in k: d[(k[5:48],r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(d.items()): print(k, len(v), 'avg', round(sum(v)/len(v),2), 'last', [round(x,1) for x in v[-6:]])
PY
done
