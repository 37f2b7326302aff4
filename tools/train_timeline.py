#!/usr/bin/env python3
"""Launch-order timeline of ONE training step from a rocprofv3 kernel trace.

  rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/train_bench.py ...
  python3 tools/train_timeline.py DIR out.txt

The step is delimited by `sgd_step_kernel` (one launch per optimiser step): the dispatches after the second-to-last and up
to the last one.  Every line: start offset (us), duration (us), idle gap before it (us), grid, workgroup, kernel.  The tail
sums time per kernel and the idle gaps -- the per-layer view tools/summarize_prof.py's per-kernel totals cannot give.
"""
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r'^void ', '', name)
    name = name.replace('yv4::', '')
    name = re.sub(r'at::native::', 'at::', name)
    return name[:110]


def main():
    src, out = sys.argv[1], sys.argv[2]
    files = glob.glob(os.path.join(src, '**', '*kernel_trace.csv'), recursive=True)
    if not files:
        sys.exit('no *kernel_trace.csv under ' + src)
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    if not rows:
        sys.exit('empty trace')
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if 'sgd_step_kernel' in r['Kernel_Name']]
    if len(marks) < 2:
        sys.exit('fewer than two optimiser steps in the trace')
    step = rows[marks[-2] + 1:marks[-1] + 1]
    t0 = int(step[0]['Start_Timestamp'])
    prev_end = t0
    per = {}
    idle = 0.0
    lines = []
    for r in step:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = max(0, s - prev_end) / 1e3
        idle += gap
        prev_end = max(prev_end, e)
        d = (e - s) / 1e3
        nm = short(r['Kernel_Name'])
        grid = '%sx%sx%s' % (r.get('Grid_Size_X', '?'), r.get('Grid_Size_Y', '?'), r.get('Grid_Size_Z', '?'))
        wg = '%s' % r.get('Workgroup_Size_X', '?')
        lines.append('%10.1f %9.1f %7.1f %16s %5s  %s' % ((s - t0) / 1e3, d, gap, grid, wg, nm))
        c = per.setdefault(nm, [0, 0.0])
        c[0] += 1
        c[1] += d
    span = (prev_end - t0) / 1e3
    busy = sum(v[1] for v in per.values())
    with open(out, 'w') as fh:
        fh.write('# one training step: %d launches, span %.1f us, kernel time %.1f us, idle gaps %.1f us\n' %
                 (len(step), span, busy, idle))
        fh.write('# %8s %9s %7s %16s %5s  kernel\n' % ('start_us', 'dur_us', 'gap_us', 'grid', 'wg'))
        fh.write('\n'.join(lines) + '\n')
        fh.write('# ---- per kernel ----\n')
        for nm, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            fh.write('# %6d %10.1f us %5.1f%%  %s\n' % (n, t, 100.0 * t / busy, nm))
    print('step: %d launches, span %.1f us, kernels %.1f us, idle %.1f us -> %s' % (len(step), span, busy, idle, out))


if __name__ == '__main__':
    main()
