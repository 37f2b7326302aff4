#!/usr/bin/env python
"""Per-launch durations of the conv kernels of the LAST bf16 inference step in a rocprofv3 kernel trace (median over the last 8
occurrences of each position in the step), wide 3x3 launches listed one by one.  Usage: w3_innet_table.py <trace_dir>"""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# step period: spacing of the last occurrences of the decode kernel
dec = [i for i, n in enumerate(names) if 'decode_filter' in n]
P = dec[-1] - dec[-2]
steps = 8
end = dec[-1] + 1
tot = collections.defaultdict(float)
cnt = collections.Counter()
per_pos = []
for pos in range(P):
    ds = []
    for s in range(steps):
        r = rows[end - (s + 1) * P + pos]
        ds.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    ds.sort()
    r = rows[end - P + pos]
    per_pos.append((r['Kernel_Name'], ds[len(ds) // 2], r.get('Grid_Size_X', r.get('Grid_Size', '?'))))
for n, d, g in per_pos:
    m = re.search(r'(conv\w+_kernel|stem_down\w*|[a-z_0-9]+_kernel)', n)
    key = m.group(1) if m else n[:40]
    tot[key] += d
    cnt[key] += 1
print(f'step period {P} launches, {sum(d for _, d, _ in per_pos):.0f} us of kernels')
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:14]:
    print(f'{k:40s} x{cnt[k]:3d} {v:8.1f} us')
print('wide 3x3 launches in step order (template args, grid, us):')
for n, d, g in per_pos:
    if 'conv3x3_wide_h16' in n:
        m = re.search(r'<(.*?)>', n)
        print(f'  {m.group(1) if m else "":20s} grid {g:>7s} {d:7.1f}')
