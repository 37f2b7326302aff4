"""How much room do the teacher-forced bounds of tests/test_gpu_zz_trajectory.py have over DIFFERENT trajectories?  The
test's trajectory is one deterministic draw; in the default (non-deterministic) mode every run of the recipe is another
draw of the same chaotic system.  REPS default-mode fp32 trajectories, the test's statistics at its snapshots, and for
every (snapshot, precision, group) the margin of the two gradient bounds -- bound minus value; negative = the test would
fail on that draw.

    python tools/teacher_forced_margin.py [--reps 3]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import test_gpu_zz_trajectory as TZ  # noqa: E402
import mmdet_yolov4_amd as pkg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=3)
    a = ap.parse_args()
    worst = dict(delta=(9.0, None), ratio=(9.0, None), loss=(9.0, None))
    for rep in range(a.reps):
        pkg.set_deterministic(False)
        losses, states = TZ.fp32_trajectory(TZ.STEPS, TZ.BATCH, TZ.SNAPS)
        pkg.set_deterministic(True)
        print(f'draw {rep}: loss at step 50 / 100 / 149: {losses[50]:.4f} {losses[100]:.4f} {losses[149]:.4f}', flush=True)
        for step in TZ.SNAPS:
            rows = {r['kind']: r for r in TZ.teacher_forced_stats(states[step], TZ.BATCH)}
            for name, loss_tol in (('fp16', 5e-3), ('bf16', 1e-2)):
                r, pr = rows[name], rows['probe_' + name]
                dpg = pr['global']['delta']
                ml = loss_tol - abs(r['loss16'] - r['loss32']) / r['loss32']
                if ml < worst['loss'][0]:
                    worst['loss'] = (ml, (rep, step, name))
                line = []
                for g, st in list(r['groups'].items()) + [('global', r['global'])]:
                    rp = pr['global']['ratio'] if g == 'global' else pr['groups'][g]['ratio']
                    md = 4 * dpg + 0.02 - st['delta']
                    mr = 2 * abs(rp - 1) + 0.15 + 0.25 * min(dpg, 1.0) - abs(st['ratio'] - 1)
                    if md < worst['delta'][0]:
                        worst['delta'] = (md, (rep, step, name, g, round(st['delta'], 3), round(dpg, 3)))
                    if mr < worst['ratio'][0]:
                        worst['ratio'] = (mr, (rep, step, name, g, round(st['ratio'], 3), round(rp, 3), round(dpg, 3)))
                    line.append((round(mr, 3), g))
                print(f'  draw {rep} step {step:3d} {name}: delta_probe {dpg:.3f} global delta {r["global"]["delta"]:.3f} '
                      f'ratio {r["global"]["ratio"]:.3f}; smallest ratio margin {min(line)}', flush=True)
    print('worst margins over all draws (bound - value):')
    for k, v in worst.items():
        print(f'  {k}: {v[0]:.4f} at {v[1]}')


if __name__ == '__main__':
    main()
