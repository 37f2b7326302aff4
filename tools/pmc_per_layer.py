#!/usr/bin/env python
"""HBM-side traffic of every conv layer of one timed step, from rocprofv3 per-dispatch counters.

Usage: pmc_per_layer.py <fetch_dir> <write_dir> <layers.json> <out.md>
The conv launches of a step are issued in plan order, so the last len(layers) conv dispatches of a profiled bench run
are the layers of its last step, in the order of the `--layers` table of the same (static) tile choice.
FETCH_SIZE / WRITE_SIZE are converted as in tools/summarize_prof.py (MI355X_MICROARCH.md, HBM section)."""
import csv
import glob
import json
import os
import sys

fetch_dir, write_dir, layers_path, out_path = sys.argv[1:5]
layers = json.load(open(layers_path))


def per_dispatch(d, counter):
    rows = []
    for f in glob.glob(os.path.join(d, '*', '*_counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter and 'conv_' in r['Kernel_Name'] and 'yv4' in r['Kernel_Name']:
                rows.append((int(r['Dispatch_Id']), r['Kernel_Name'], float(r['Counter_Value'])))
    rows.sort()
    assert len(rows) >= len(layers), (len(rows), len(layers))
    return rows[-len(layers):]


fetch = per_dispatch(fetch_dir, 'FETCH_SIZE')
write = per_dispatch(write_dir, 'WRITE_SIZE')
N = int(os.environ.get('YV4_BATCH', '32'))
ES = int(os.environ.get('YV4_ESIZE', '4'))                 # bytes per activation / weight element (2 on the 16-bit path)
lines = ['| # | layer | tile | µs | algorithmic MB (in + w + out + residual) | fetched MB | written MB | traffic ÷ algorithmic |',
         '|---|---|---|---|---|---|---|---|']
tot_a = tot_t = 0.0
by_tile = {}
for i, (L, f, w) in enumerate(zip(layers, fetch, write)):
    tile = L['tile'].replace('dma', '').replace('h16_', '')
    want = 'stem' if 'stem' in tile else ' %s,' % tile.replace('x', ', ')           # '<128, 64,' or '<true, 128, 64,'
    alt = 'stem' if 'stem' in tile else 'Li%sELi%sE' % tuple(tile.split('x'))      # mangled spelling of the same template
    assert (want in f[1] or alt in f[1]) and (want in w[1] or alt in w[1]), (i, L, f[1])   # launch order, checked per row
    Ho, Wo = -(-L['H'] // L['stride']), -(-L['W'] // L['stride'])
    es = ES
    ies = 4 if 'stem' in tile else es                                              # the image stays fp32 on every path
    a_in, a_w, a_out = N * L['H'] * L['W'] * L['Cin'] * ies, L['Cout'] * L['Cin'] * L['k'] ** 2 * ies, N * Ho * Wo * L['Cout'] * es
    a_res = a_out if L.get('residual') else 0
    alg = a_in + a_w + a_out + a_res
    rd, wr = f[2] * 1024 * 2, w[2] * 1024
    tot_a += alg
    tot_t += rd + wr
    t = by_tile.setdefault(L['tile'], [0.0, 0.0, 0])
    t[0] += alg; t[1] += rd + wr; t[2] += 1
    lines.append('| %d | %dx%d s%d %d->%d @%dx%d | %s | %.0f | %.1f (%.1f + %.1f + %.1f + %.1f) | %.1f | %.1f | %.2f |'
                 % (i, L['k'], L['k'], L['stride'], L['Cin'], L['Cout'], L['H'], L['W'], L['tile'], L['us'], alg / 1e6,
                    a_in / 1e6, a_w / 1e6, a_out / 1e6, a_res / 1e6, rd / 1e6, wr / 1e6, (rd + wr) / alg))
head = ['# Per-layer HBM-side traffic of one step (YOLOv4-L 608x608 %s, batch %d)' % ('fp32' if ES == 4 else '16-bit', N), '',
        'Counters: `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) over `bench.py`; per-dispatch values of the',
        'last step joined to the layer table by launch order (kernel template checked per row).',
        'Whole step: %.0f MB algorithmic, %.0f MB counted = %.2fx.' % (tot_a / 1e6, tot_t / 1e6, tot_t / tot_a), '']
for k, (a, t, n) in sorted(by_tile.items()):
    head.append('* %s: %d layers, %.0f MB algorithmic, %.0f MB counted = %.2fx' % (k, n, a / 1e6, t / 1e6, t / a))
open(out_path, 'w').write('\n'.join(head + [''] + lines) + '\n')
print('\n'.join(head))
