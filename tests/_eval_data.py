"""Shared helpers of the evaluation tests: the dataset stored in tests/golden/eval.npz and seeded problems."""
import numpy as np

THRS10 = [0.5 + 0.05 * x for x in range(10)]
SCALES = dict(Scale_S=(0, 32), Scale_M=(32, 96), Scale_L=(96, 10000))
REPORT = [('map', lambda x: x['breakdown'] == 'All'),
          ('map50', lambda x: x['iou_threshold'] == 0.5 and x['breakdown'] == 'All'),
          ('map75', lambda x: x['iou_threshold'] == 0.75 and x['breakdown'] == 'All'),
          ('s_map', lambda x: x['breakdown'] == 'Scale_S'),
          ('m_map', lambda x: x['breakdown'] == 'Scale_M'),
          ('l_map', lambda x: x['breakdown'] == 'Scale_L')]


def dataset(z):
    ni, nc = int(z['ds/num_img']), int(z['ds/num_cls'])
    dets = [[z[f'ds/det/{i}/{c}'] for c in range(nc)] for i in range(ni)]
    annos = []
    for i in range(ni):
        attrs = {k: z[f'ds/attr/{k}/{i}'] for k in ('iscrowd', 'ignore') if f'ds/attr/{k}/{i}' in z.files}
        annos.append(dict(gt_bboxes=z[f'ds/gt_bboxes/{i}'], gt_labels=z[f'ds/gt_labels/{i}'], gt_attrs=attrs))
    return dets, annos, [f'c{i}' for i in range(nc)]


def result_table(res, classes):
    order = ['All'] + list(SCALES)
    key = np.array([[classes.index(k['class_name']), order.index(k['breakdown']),
                     int(round((k['iou_threshold'] - 0.5) / 0.05)), v['num_det'], v['num_gt']] for k, v in res],
                   np.int64)
    return key, np.array([v['recall'] for _, v in res], np.float64), np.array([v['mAP'] for _, v in res], np.float32)


def random_problem(rng, nd, ng, degenerate=False):
    def boxes(n):
        xy = rng.uniform(0, 100, (n, 2))
        wh = rng.uniform(0, 60, (n, 2))
        return np.concatenate([xy, xy + wh], 1).astype(np.float32)
    d, g = boxes(nd), boxes(ng)
    n = min(nd, ng)
    d[:n] = g[:n] + rng.normal(0, 3, (n, 4)).astype(np.float32)
    if degenerate and n:
        d[0] = g[0]
        d[-1, 2:] = d[-1, :2]
        g[-1, 2:] = g[-1, :2]
    return d, g, rng.random(ng) < 0.3, rng.random(ng) < 0.3
