import csv,glob,sys,collections,re
d=sys.argv[1]; pat=sys.argv[2]
f=glob.glob(d+'/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if pat in r['Kernel_Name']]
print(rows[0].keys() if rows else 'none')
agg=collections.defaultdict(list)
for r in rows:
    key=(r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Grid_Size_Y'), r.get('LDS_Block_Size') or r.get('LDS_Block_Size_v'))
    agg[key].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
tot=0
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    print(k, len(v), f'avg {sum(v)/len(v)/1e3:7.1f} us  total {sum(v)/1e6:7.2f} ms')
    tot+=sum(v)
print('total', tot/1e6)
