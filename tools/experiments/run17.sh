cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for c in "VALUBusy MemUnitBusy" "LdsUtil MemUnitStalled" "MeanOccupancyPerCU LDSBankConflict"; do
  n=$(echo $c | tr ' ' '_'); rm -rf gpurun_out/pmcdw_$n
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmcdw_$n -o p -- python bench.py --steps 1 --warmup 1 --no-kernel-probe --no-cpu-baseline --batch 512 > gpurun_out/pmcdw_$n.log 2>&1
  python - "$n" <<'PY'
import csv,glob,sys,collections,re
n=sys.argv[1]
fs=glob.glob(f'gpurun_out/pmcdw_{n}/*counter_collection.csv')
if not fs: print(n,'no output'); sys.exit()
d=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    k=re.sub(r'\(.*','',r['Kernel_Name']).replace('void ','')[:44]
    d[(k,r['Counter_Name'])].append(float(r['Counter_Value']))
keep=('dwf_','conv_gemm','gemm_tn','attn_','ln_','bn_apply','bn_bwd_apply','colreduce')
for k,v in sorted(d.items()):
    if any(x in k[0] for x in keep): print(f'{k[0]:46s}{k[1]:22s} n={len(v):4d} avg {sum(v)/len(v):8.2f}')
PY
done
