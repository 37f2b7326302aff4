import csv,glob,sys,collections,re
d=sys.argv[1]
f=glob.glob(d+'/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# steady part: last 60% of the rows
n=len(rows); rows=rows[int(n*0.4):]
span=int(rows[-1]['End_Timestamp'])-int(rows[0]['Start_Timestamp'])
ks=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows)
gaps=0; prev_end=None; gap_after=collections.Counter(); cnt=collections.Counter(); dur=collections.Counter()
def short(nm):
    nm=re.sub(r'\(.*','',nm); nm=nm.replace('void ','')
    return nm[:70]
for r in rows:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    nm=short(r['Kernel_Name'])
    if prev_end is not None and s>prev_end:
        gaps+=s-prev_end; gap_after[prev_nm]+=s-prev_end
    cnt[nm]+=1; dur[nm]+=e-s
    prev_end=max(prev_end or 0,e); prev_nm=nm
print(f'kernels {len(rows)} span {span/1e6:.2f} ms kernel-sum {ks/1e6:.2f} ms gaps {gaps/1e6:.2f} ms')
for nm,v in dur.most_common(int(sys.argv[2]) if len(sys.argv)>2 else 12):
    print(f'  {nm:72s} n={cnt[nm]:5d} dur {v/1e6:8.2f} ms  avg {v/cnt[nm]/1e3:7.1f} us  gap-after avg {gap_after[nm]/cnt[nm]/1e3:6.2f} us')
