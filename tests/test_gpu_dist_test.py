"""The distributed test loop end to end: two processes share cuda:0 (gloo group -- the only 2-rank transport a
1-GPU box offers), each runs ``multi_gpu_test`` over its ``DistributedSampler``-ordered share of a 5-image
dataset through the fused input pipeline and the HIP detector, and rank 0 must hold exactly the list one process
gets from ``single_gpu_test`` over the whole dataset (mmdet/apis/test.py:16-113, tools/test.py:185-196).
Bit-exact: same kernels on the same images, only the order of evaluation differs."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import json, os, sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch
    import torch.distributed as dist
    import mmdet_yolov4_amd as pkg
    from mmdet_yolov4_amd import dist as D
    rank = int(os.environ['RANK'])
    dist.init_process_group('gloo', rank=rank, world_size=2)
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    det = pkg.build_detector(dict(
        type='SingleStageDetector',
        backbone=dict(type='DarknetCSP', scale=[['conv', 'bottleneck', 'csp', 'csp'], [None, 1, 1, 1], [8, 16, 16, 32]],
                      out_indices=[1, 2, 3]),
        neck=dict(type='YOLOV4Neck', in_channels=[16, 16, 32], out_channels=[16, 16, 32], csp_repetition=1),
        bbox_head=dict(type='YOLOCSPHead', num_classes=3, in_channels=[16, 16, 32], featmap_strides=[4, 8, 16],
                       anchor_generator=dict(type='YOLOV4AnchorGenerator', strides=[4, 8, 16],
                                             base_sizes=[[(8, 8)] * 3, [(16, 16)] * 3, [(32, 32)] * 3])),
        test_cfg=dict(nms_pre=-1, score_thr=0.001, nms=dict(type='nms', iou_threshold=0.65), max_per_img=10)))
    det.init_weights()
    det.eval().to(dev)
    rng = np.random.default_rng(7)
    size = 5
    images = [rng.integers(0, 256, (90 + 10 * i, 140 - 6 * i, 3), dtype=np.uint8) for i in range(size)]
    pipe = pkg.FusedTestPipeline(img_scale=(128, 128), device=dev)

    def loader(indices):                                  # samples_per_gpu=1, the collate's nesting per augmentation
        for i in indices:
            batch, metas = pipe([images[i]])
            yield dict(img=[batch], img_metas=[metas])

    mine = D.sampler_indices(size, rank, 2)
    merged = pkg.multi_gpu_test(det, loader(mine), size=size, gpu_collect=False)
    out = dict(rank=rank, none=merged is None)
    if rank == 0:
        whole = pkg.single_gpu_test(det, loader(range(size)))
        out['n'] = len(merged)
        out['dets'] = int(sum(len(c) for r in whole for c in r))
        out['same'] = all(np.array_equal(a, b) for ra, rb in zip(merged, whole) for a, b in zip(ra, rb))
    print('RESULT ' + json.dumps(out), flush=True)
    dist.destroy_process_group()
''')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_test_loop_matches_one_process(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % ROOT)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK='0', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), OMP_NUM_THREADS='1')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdin=subprocess.DEVNULL,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out
        outs.append(json.loads([l for l in out.splitlines() if l.startswith('RESULT ')][0][7:]))
    outs.sort(key=lambda o: o['rank'])
    assert outs[1]['none'] is True and outs[0]['none'] is False
    assert outs[0]['n'] == 5 and outs[0]['dets'] > 0
    assert outs[0]['same'] is True
