#!/usr/bin/env python
"""Condense rocprofv3 CSVs into small summaries.  Usage: summarize_prof.py <raw_dir> <out_dir>"""
import collections
import csv
import glob
import json
import os
import re
import sys

R03 = '--r03' in sys.argv      # (the layout of rounds 3-5: one directory per profiled command)
argv = [a for a in sys.argv[1:] if a != '--r03']
raw, out = argv[0], argv[1]
os.makedirs(out, exist_ok=True)


def short(name):
    name = re.sub(r'^void ', '', name)
    m = re.match(r'_ZN3yv4(\d+)([A-Za-z0-9_]+?)I(DF16b|DF16_|f)E', name)     # templated on __bf16 / _Float16
    if m:
        base = m.group(2)[:int(m.group(1))]
        return f"yv4::{base}<{ {'DF16b': '__bf16', 'DF16_': '_Float16', 'f': 'float'}[m.group(3)] }>"
    m = re.match(r'(yv4::[A-Za-z0-9_]+(<[^>]*>)?)', name)
    if m:
        return m.group(1)
    return name.split('(')[0][:70]


def timed_region_stats(trace_csv, steps):
    """Per-kernel statistics over the TIMED steps only.  rocprofv3 --stats covers the whole process: model build, BatchNorm
    calibration in fp32, warm-up -- in the 16-bit runs the fp32 calibration plan's kernels headed the table and made its
    Percentage column meaningless.  The timed region is the last `steps` repetitions of one launch sequence: the period is
    found as the smallest P for which the last steps * P kernel names are P-periodic, and the statistics are taken over
    exactly those launches.  Returns (rows, period) or (None, 0) when no period is found."""
    rows = list(csv.DictReader(open(trace_csv)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    names = [r['Kernel_Name'] for r in rows]
    n = len(names)
    # candidate periods: the spacing of any kernel whose last occurrences are equally spaced `steps` times in a row
    occ = collections.defaultdict(list)
    for i, nm in enumerate(names):
        occ[nm].append(i)
    cands = set()
    for o in occ.values():
        d = [o[i + 1] - o[i] for i in range(len(o) - 1)]
        run = 1
        for i in range(1, len(d)):
            run = run + 1 if (d[i] == d[i - 1] and d[i] >= 20) else 1
            if run >= steps - 1:
                cands.add(d[i])
    # the LAST stretch of steps * P launches that is P-periodic (what follows it -- instrumented steps with event records,
    # read-backs -- is not part of the timed region); the smallest such period
    best = None
    for P in sorted(cands):
        for e in range(n, steps * P - 1, -1):
            seg = names[e - steps * P:e]
            if all(seg[i] == seg[i - P] for i in range(P, steps * P)):
                if best is None or e > best[0]:
                    best = (e, P)
                break
    if best is None:
        return None, 0
    end, period = best
    n = end
    agg = collections.OrderedDict()
    for r in rows[end - steps * period:end]:
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        a = agg.setdefault(r['Kernel_Name'], [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    out = [[short(k), a[0], a[1], round(a[1] / a[0], 1), round(100.0 * a[1] / tot, 3), a[2], a[3]]
           for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])]
    return out, period


def marker_region_stats(trace_csv, steps, marker):
    """Per-kernel statistics over the last `steps` steps of a trace whose launch ORDER is not periodic (the train step with
    its weight gradients on a side stream: the two streams interleave differently every step).  `marker` names a kernel
    that runs exactly once per step on the main stream (the optimizer's yv4_sgd_step): the region is everything launched
    after the end of the (steps + 1)-th last marker up to the end of the last one."""
    rows = list(csv.DictReader(open(trace_csv)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [r for r in rows if marker in r['Kernel_Name']]
    if len(marks) < steps + 1:
        return None, 0
    t0, t1 = int(marks[-steps - 1]['End_Timestamp']), int(marks[-1]['End_Timestamp'])
    agg = collections.OrderedDict()
    n = 0
    for r in rows:
        if not (t0 < int(r['Start_Timestamp']) and int(r['End_Timestamp']) <= t1 + 1):
            continue
        n += 1
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        a = agg.setdefault(r['Kernel_Name'], [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    out = [[short(k), a[0], a[1], round(a[1] / a[0], 1), round(100.0 * a[1] / tot, 3), a[2], a[3]]
           for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])]
    return out, (n // steps, (t1 - t0) / steps / 1e6)


def pmc_of(run_dir):
    """{kernel: mean counters per launch} of one profiled command (its pmc_fetch / pmc_write passes)."""
    pm = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in ('pmc_fetch', 'pmc_write'):
        for f in glob.glob(os.path.join(run_dir, d, '*', '*_counter_collection.csv')):
            for r in csv.DictReader(open(f)):
                if 'yv4' not in r['Kernel_Name']:
                    continue
                pm[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    res = {}
    for k, cs in pm.items():
        e = {c: sum(v) / len(v) for c, v in cs.items()}
        e['launches_sampled'] = max(len(v) for v in cs.values())
        if 'FETCH_SIZE' in e:      # MI355X_MICROARCH.md, HBM: FETCH_SIZE (KiB) tallies 64 B per 128-B request on gfx950 -> x2
            e['hbm_read_bytes_per_launch'] = e['FETCH_SIZE'] * 1024 * 2
        if 'WRITE_SIZE' in e:
            e['hbm_write_bytes_per_launch'] = e['WRITE_SIZE'] * 1024
        if 'hbm_read_bytes_per_launch' in e and 'hbm_write_bytes_per_launch' in e:
            e['hbm_bytes_per_launch'] = e['hbm_read_bytes_per_launch'] + e['hbm_write_bytes_per_launch']
        res[k] = e
    return res


if R03:
    # one directory per profiled command: <name>/{trace,pmc_fetch,pmc_write,bench_profiled.json,bench_pmc.json}
    merged = {'f32': {}, 'h16': {}}
    for run_dir in sorted(glob.glob(os.path.join(raw, '*'))):
        name = os.path.basename(run_dir)
        if not os.path.isdir(os.path.join(run_dir, 'trace')):
            continue
        steps_file = os.path.join(run_dir, 'steps.txt')
        done = False
        marker_file = os.path.join(run_dir, 'marker.txt')
        if os.path.exists(steps_file) and os.path.exists(marker_file):
            steps = int(open(steps_file).read().split()[0])
            for f in glob.glob(os.path.join(run_dir, 'trace', '*', '*_kernel_trace.csv')):
                rows, info = marker_region_stats(f, steps, open(marker_file).read().strip())
                if rows:
                    with open(os.path.join(out, name + '_kernel_stats.csv'), 'w', newline='') as g:
                        g.write(f'# timed region only: the last {steps} steps, cut at the once-per-step marker kernel (the launch order of two '
                                f'streams is not periodic): {info[0]} launches and {info[1]:.2f} ms of wall time per step; kernel times of both '
                                f'streams are summed, so the column total exceeds the wall time where they overlap\n')
                        w = csv.writer(g)
                        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
                        w.writerows(rows)
                    done = True
        elif os.path.exists(steps_file):
            steps = int(open(steps_file).read().split()[0])
            for f in glob.glob(os.path.join(run_dir, 'trace', '*', '*_kernel_trace.csv')):
                rows, period = timed_region_stats(f, steps)
                if rows:
                    with open(os.path.join(out, name + '_kernel_stats.csv'), 'w', newline='') as g:
                        g.write(f'# timed region only: the last {steps} steps x {period} launches per step of the kernel trace '
                                f'(calibration, warm-up and set-up launches excluded)\n')
                        w = csv.writer(g)
                        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
                        w.writerows(rows)
                    done = True
        for f in ([] if done else glob.glob(os.path.join(run_dir, 'trace', '*', '*_kernel_stats.csv'))):
            rows = list(csv.DictReader(open(f)))
            with open(os.path.join(out, name + '_kernel_stats.csv'), 'w', newline='') as g:
                g.write('# whole process (no step period found in the trace): includes set-up, calibration and warm-up launches\n')
                w = csv.writer(g)
                w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
                for r in rows:
                    w.writerow([short(r['Name']), r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'],
                                r['MinNs'], r['MaxNs']])
        res = pmc_of(run_dir)
        run = {}
        for tag in ('bench_pmc.json', 'bench_profiled.json'):
            p = os.path.join(run_dir, tag)
            if os.path.exists(p) and os.path.getsize(p) > 2:
                line = json.load(open(p))
                if tag == 'bench_profiled.json':
                    json.dump(line, open(os.path.join(out, name + '_bench_profiled.json'), 'w'))
                rf = line.get('roofline') or {}
                if not run and rf.get('tiles'):
                    # the launch set of the profiled command: bench.py refuses this summary for a run whose own differs
                    run = dict(tiles=rf['tiles'], run_key=rf.get('run_key'), bench_line_of='pmc FETCH_SIZE pass'
                               if tag == 'bench_pmc.json' else 'kernel-trace pass')
        res['_run'] = run
        merged['f32' if name.endswith('_f32') else 'h16'][name] = res
    if merged['f32']:
        json.dump(merged['f32'], open(os.path.join(out, 'pmc_per_kernel.json'), 'w'), indent=1, sort_keys=True)
    if merged['h16']:
        json.dump(merged['h16'], open(os.path.join(out, 'pmc_per_kernel_h16.json'), 'w'), indent=1, sort_keys=True)
    print('summary written to', out)
    sys.exit(0)

# 1. kernel stats (inference bench, training step)
for sub, dst in (('trace', 'kernel_stats.csv'), ('trace_train', 'train_kernel_stats.csv'),
                 ('trace_train_bf16', 'train_bf16_kernel_stats.csv')):
  for f in glob.glob(os.path.join(raw, sub, '*', '*_kernel_stats.csv')):
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(out, dst), 'w', newline='') as g:
        w = csv.writer(g)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
        for r in rows:
            w.writerow([short(r['Name']), r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'],
                        r['MinNs'], r['MaxNs']])

# 2. PMC per kernel (mean per launch)
pm = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('pmc_fetch', 'pmc_write', 'pmc_sq'):
    for f in glob.glob(os.path.join(raw, d, '*', '*_counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            if 'yv4::' not in r['Kernel_Name']:
                continue
            pm[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
res = {}
for k, cs in pm.items():
    e = {c: sum(v) / len(v) for c, v in cs.items()}
    e['launches_sampled'] = max(len(v) for v in cs.values())
    if 'FETCH_SIZE' in e:
        # MI355X_MICROARCH.md, HBM: FETCH_SIZE (KiB) counts 64 B per 128-B request on gfx950 -> x2
        e['hbm_read_bytes_per_launch'] = e['FETCH_SIZE'] * 1024 * 2
    if 'WRITE_SIZE' in e:
        e['hbm_write_bytes_per_launch'] = e['WRITE_SIZE'] * 1024
    if 'hbm_read_bytes_per_launch' in e and 'hbm_write_bytes_per_launch' in e:
        e['hbm_bytes_per_launch'] = e['hbm_read_bytes_per_launch'] + e['hbm_write_bytes_per_launch']
    if 'TCC_HIT_sum' in e:
        e['l2_hit_rate'] = e['TCC_HIT_sum'] / max(e['TCC_HIT_sum'] + e['TCC_MISS_sum'], 1)
    res[k] = e
json.dump(res, open(os.path.join(out, 'pmc_per_kernel.json'), 'w'), indent=1, sort_keys=True)
for name in ('conv_shapes.txt', 'bench.json', 'layers.json', 'train_bench.json', 'bench_bf16.json', 'layers_bf16.json',
             'train_bench_bf16.json', 'conv_shapes_bf16.txt'):
    p = os.path.join(raw, name)
    if os.path.exists(p):
        open(os.path.join(out, name), 'w').write(open(p).read())
print('summary written to', out)
