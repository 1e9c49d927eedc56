import csv,re,collections,sys
d=sys.argv[1]
import glob
rows=list(csv.DictReader(open(glob.glob(f'{d}/*kernel_trace.csv')[0])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'stem_fwd_kernel' in r['Kernel_Name'] or 'nchw_to_nhwc' in r['Kernel_Name']]
step=rows[idx[-2]:idx[-1]]
tot=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in step)
print(len(step),'kernels', tot/1e6,'ms/step')
def short(n):
    n=re.sub(r"^void ","",n).replace('(anonymous namespace)::',''); return re.sub(r"\(.*","",n)[:48]
byname=collections.defaultdict(lambda:[0,0])
agg=collections.defaultdict(lambda:[0,0])
for r in step:
    dd=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    byname[short(r['Kernel_Name'])][0]+=dd; byname[short(r['Kernel_Name'])][1]+=1
    key=(short(r['Kernel_Name']), r['Grid_Size_X'], r['Grid_Size_Y'], r['Workgroup_Size_X'])
    agg[key][0]+=dd; agg[key][1]+=1
print("--- by kernel")
for k,v in sorted(byname.items(), key=lambda kv:-kv[1][0])[:int(sys.argv[2]) if len(sys.argv)>2 else 22]:
    print(f"{v[0]/tot*100:6.2f}% {v[0]/1e3:8.1f}us n={v[1]:3d}  {k}")
print("--- by kernel+grid")
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][0])[:int(sys.argv[3]) if len(sys.argv)>3 else 25]:
    print(f"{v[0]/1e3:8.1f}us n={v[1]:3d} avg={v[0]/v[1]/1e3:8.1f}us grid=({int(k[1])//int(k[3])},{k[2]})x{k[3]} {k[0]}")
