"""SyncBN (norm_cfg type 'SyncBN' of configs/yolov5_ddp = torch.nn.SyncBatchNorm) on the HIP BN kernels:
two processes share cuda:0 and exchange the statistics over a gloo group (the only 2-rank transport a
1-GPU box offers; on a node the same calls run over RCCL).  Each rank checks its outputs against the
definition computed in float64 on the CPU: batch statistics over BOTH ranks' rows, dx with the global
sums, LOCAL dgamma / dbeta (what torch.nn.SyncBatchNorm returns), running statistics with the unbiased
variance over the total row count.  Tolerance 2e-5 (fp32 elementwise arithmetic, double reductions)."""
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
    import torch, torch.nn as nn, torch.nn.functional as F
    import torch.distributed as dist
    import mmdet_yolov4_amd as pkg
    from mmdet_yolov4_amd import train_ops as T
    rank = int(os.environ['RANK'])
    dist.init_process_group('gloo', rank=rank, world_size=2)
    dev = torch.device('cuda', 0)
    C, H, W = 16, 6, 5
    sizes = [2, 3]                                      # ragged: the ranks hold different row counts
    g = torch.Generator().manual_seed(7)
    xs = [torch.randn(n, C, H, W, generator=g) * 2 + 0.5 for n in sizes]
    ws = [torch.randn(n, C, H, W, generator=g) for n in sizes]
    gamma0 = torch.rand(C, generator=g) + 0.5
    beta0 = torch.randn(C, generator=g) * 0.2
    out = {}
    for dtype, tol in ((torch.float32, 2e-5), (torch.bfloat16, 3e-2)):
        for act in ((0, 0.0), (1, 0.0), (2, 0.1)):          # YV4_ACT_NONE / MISH / LEAKY
            bn = nn.SyncBatchNorm(C, eps=1e-3, momentum=0.03).to(dev).train()
            with torch.no_grad():
                bn.weight.copy_(gamma0); bn.bias.copy_(beta0)
                bn.running_mean.fill_(0.25)
                bn.running_var.fill_(1.5)
            x = xs[rank].to(dev).to(dtype).requires_grad_(True)
            y = T.bn_act(x, bn, act)
            (y.float() * ws[rank].to(dev)).sum().backward()
            # ---- the definition, float64 on the CPU, over both ranks' rows ---------------------------
            xq = [t.to(dtype).double() for t in xs]
            cat = torch.cat(xq, 0)
            mean = cat.mean((0, 2, 3)); var = cat.var((0, 2, 3), unbiased=False)
            n_tot = cat.numel() // C
            gam = [gamma0.double().clone().requires_grad_(True) for _ in sizes]
            bet = [beta0.double().clone().requires_grad_(True) for _ in sizes]
            leaves = [t.clone().requires_grad_(True) for t in xq]
            allx = torch.cat(leaves, 0)
            m = allx.mean((0, 2, 3), keepdim=True); v = allx.var((0, 2, 3), unbiased=False, keepdim=True)
            tot = 0
            ys = []
            for r, n in enumerate(sizes):
                lo = sum(sizes[:r])
                xhat = (allx[lo:lo + n] - m) / torch.sqrt(v + 1e-3)
                z = xhat * gam[r].view(1, -1, 1, 1) + bet[r].view(1, -1, 1, 1)
                z = {0: z, 2: F.leaky_relu(z, 0.1), 1: z * torch.tanh(F.softplus(z))}[act[0]]
                ys.append(z)
                tot = tot + (z * ws[r].double()).sum()
            tot.backward()
            err = lambda a, b: float((a.detach().double().cpu() - b.detach()).abs().max() / (b.detach().abs().max() + 1e-12))
            e = dict(y=err(y, ys[rank]), dx=err(x.grad, leaves[rank].grad), dgamma=err(bn.weight.grad, gam[rank].grad),
                     dbeta=err(bn.bias.grad, bet[rank].grad),
                     rmean=err(bn.running_mean, 0.97 * 0.25 + 0.03 * mean),
                     rvar=err(bn.running_var, 0.97 * 1.5 + 0.03 * var * n_tot / (n_tot - 1)),
                     tracked=int(bn.num_batches_tracked))
            out[str(dtype) + str(act[0])] = dict(e, tol=tol)
    print('RESULT ' + json.dumps(dict(rank=rank, out=out)), flush=True)
    dist.destroy_process_group()
''')


def test_syncbn_two_ranks_one_gpu(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % ROOT)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK='0', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), OMP_NUM_THREADS='1')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out
        res = json.loads([l for l in out.splitlines() if l.startswith('RESULT ')][0][7:])
        for name, e in res['out'].items():
            assert e['tracked'] == 1
            for k in ('y', 'dx', 'dgamma', 'dbeta'):
                assert e[k] < e['tol'], (res['rank'], name, k, e)
            for k in ('rmean', 'rvar'):
                assert e[k] < 2e-5 if 'float32' in name else e[k] < 1e-3, (res['rank'], name, k, e)


def test_syncbn_single_process_is_plain_bn():
    """world_size 1 / no process group: SyncBatchNorm takes the local-statistics kernels (torch does too)."""
    import torch
    import torch.nn as nn
    from mmdet_yolov4_amd import train_ops as T
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    x = torch.randn(3, 8, 5, 7, device=dev)
    a, b = nn.SyncBatchNorm(8).to(dev).train(), nn.BatchNorm2d(8).to(dev).train()
    ya, yb = T.bn_act(x, a, (1, 0.0)), T.bn_act(x, b, (1, 0.0))
    assert torch.equal(ya, yb) and torch.equal(a.running_var, b.running_var)
