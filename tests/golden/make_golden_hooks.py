#!/usr/bin/env python
"""Generate tests/golden/hooks.npz by running the REFERENCE's three training hooks
(/root/reference/mmdet/core/custom_hooks/*.py, imported from where they lie) on a toy model.

The hooks' base classes come from mmcv (third-party, absent): ``Hook`` (empty stage methods),
``HOOKS`` (a registry) and ``Fp16OptimizerHook`` are restated below from mmcv 1.3.x -- only what
the reference's subclasses call: the GradScaler construction, ``clip_grads`` =
``clip_grad_norm_`` over parameters that have gradients.  On this CPU-only container
``torch.cuda.amp.GradScaler`` is disabled (scale 1, ``step`` = ``optimizer.step``), so the fixture
pins: accumulation boundaries, summed (not averaged) gradients, clipping, SGD-Nesterov with one
group per parameter, the warm-up lr/momentum sequences, EMA momentum/updates and the epoch swaps.

Run in the build container only; the GPU box never sees /root/reference.
"""
import importlib
import json
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF = '/root/reference'


def install_hook_shim():
    class Hook:
        def before_run(self, runner): pass
        def after_run(self, runner): pass
        def before_epoch(self, runner): pass
        def after_epoch(self, runner): pass
        def before_iter(self, runner): pass
        def after_iter(self, runner): pass
        def before_train_epoch(self, runner): self.before_epoch(runner)
        def after_train_epoch(self, runner): self.after_epoch(runner)
        def before_train_iter(self, runner): self.before_iter(runner)
        def after_train_iter(self, runner): self.after_iter(runner)

    class _Reg:
        def __init__(self):
            self.d = {}

        def register_module(self, name=None, force=False, module=None):
            def reg(cls):
                self.d[name or cls.__name__] = cls
                return cls
            return reg(module) if module is not None else reg

    class Fp16OptimizerHook(Hook):
        """mmcv 1.3.x runner/hooks/optimizer.py (torch >= 1.6 branch), constructor + clip_grads."""

        def __init__(self, grad_clip=None, coalesce=True, bucket_size_mb=-1, loss_scale=512., distributed=True):
            from torch.cuda.amp import GradScaler
            self.grad_clip = grad_clip
            self.coalesce = coalesce
            self.bucket_size_mb = bucket_size_mb
            self.distributed = distributed
            self._scale_update_param = None
            if loss_scale == 'dynamic':
                self.loss_scaler = GradScaler()
            elif isinstance(loss_scale, float):
                self._scale_update_param = loss_scale
                self.loss_scaler = GradScaler(init_scale=loss_scale)
            elif isinstance(loss_scale, dict):
                self.loss_scaler = GradScaler(**loss_scale)
            else:
                raise ValueError(loss_scale)

        def clip_grads(self, params):
            params = list(filter(lambda p: p.requires_grad and p.grad is not None, params))
            if len(params) > 0:
                return torch.nn.utils.clip_grad_norm_(params, **self.grad_clip)

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    hooks = _Reg()
    mod('mmcv')
    mod('mmcv.runner', HOOKS=hooks, Hook=Hook, Fp16OptimizerHook=Fp16OptimizerHook)
    mod('mmcv.runner.dist_utils', allreduce_grads=lambda *a, **k: None, get_dist_info=lambda: (0, 1))
    mod('mmcv.utils', TORCH_VERSION=torch.__version__)
    mod('mmcv.parallel', is_module_wrapper=lambda m: False)
    for name, path in (('mmdet', 'mmdet'), ('mmdet.core', 'mmdet/core'),
                       ('mmdet.core.custom_hooks', 'mmdet/core/custom_hooks')):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, path)]
        sys.modules[name] = m
    imp = importlib.import_module
    return types.SimpleNamespace(accum=imp('mmdet.core.custom_hooks.accum_optim_hooks'),
                                 ema=imp('mmdet.core.custom_hooks.ema_hooks'),
                                 warm=imp('mmdet.core.custom_hooks.warmup_hooks'))


from toy_model import Toy, toy_batches, toy_groups  # noqa: E402  (fixture model shared with the tests)


CFG = dict(lr=0.01, momentum=0.937, weight_decay=0.0005, nesterov=True, max_norm=2.0, nominal_batch_size=8,
           samples_per_gpu=4, warmup_iters=8, lr_weight_warmup_ratio=0., lr_bias_warmup_ratio=10.,
           momentum_warmup_ratio=0.95, ema_momentum=0.9, ema_warm_up=3, epochs=2, iters_per_epoch=6, seed=5)


def main():
    if not os.path.isdir(REF):
        print('reference not present: nothing to do')
        return
    ref = install_hook_shim()
    torch.manual_seed(0)
    c = CFG
    model = Toy()
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * 0.3)
        model.bn.weight.add_(1.0)
    data = {f'init/{k}': v.detach().clone().numpy() for k, v in model.state_dict().items()}
    opt = torch.optim.SGD(toy_groups(model, c['lr'], c['weight_decay']), lr=c['lr'], momentum=c['momentum'],
                          weight_decay=c['weight_decay'], nesterov=c['nesterov'])

    class Sampler:
        samples_per_gpu = c['samples_per_gpu']

    class Loader(list):
        sampler = Sampler()

    class Log:
        def __init__(self): self.rows = []
        def update(self, d, n): self.rows.append(d)

    class Logger:
        def warning(self, *a): raise AssertionError(a)

    runner = types.SimpleNamespace(model=model, optimizer=opt, iter=0, epoch=0, outputs=None, logger=Logger(),
                                   log_buffer=Log(), meta={'config': 'resume_from = None\n'},
                                   data_loader=Loader(toy_batches(c['iters_per_epoch'], c['samples_per_gpu'], c['seed'])))
    runner.resume = lambda ckpt: None
    hooks = [  # priority order of the recipe: EMA 'HIGH' before the NORMAL ones (registration order kept)
        ref.ema.StateEMAHook(momentum=c['ema_momentum'], nominal_batch_size=c['nominal_batch_size'],
                             warm_up=c['ema_warm_up'], resume_from=None),
        ref.accum.Fp16GradAccumulateOptimizerHook(nominal_batch_size=c['nominal_batch_size'],
                                                  grad_clip=dict(max_norm=c['max_norm'], norm_type=2),
                                                  loss_scale='dynamic'),
        ref.warm.DetailedLinearWarmUpHook(warmup_iters=c['warmup_iters'],
                                          lr_weight_warmup_ratio=c['lr_weight_warmup_ratio'],
                                          lr_bias_warmup_ratio=c['lr_bias_warmup_ratio'],
                                          momentum_warmup_ratio=c['momentum_warmup_ratio']),
    ]

    def call(stage):
        for h in hooks:
            getattr(h, stage)(runner)

    names = [n for n, _ in model.named_parameters()]
    call('before_run')
    lrs, moms, losses = [], [], []
    for ep in range(c['epochs']):
        model.train()
        call('before_train_epoch')
        data[f'swap_in/{ep}/conv.weight'] = model.conv.weight.detach().clone().numpy()
        for batch in runner.data_loader:
            call('before_train_iter')
            lrs.append([g['lr'] for g in opt.param_groups])
            moms.append([g['momentum'] for g in opt.param_groups])
            runner.outputs = model.train_step(batch, opt)
            losses.append(float(runner.outputs['loss']))
            call('after_train_iter')
            it = runner.iter
            for k, v in model.state_dict().items():
                data[f'iter{it}/{k}'] = v.detach().clone().float().numpy()
            runner.iter += 1
        call('after_train_epoch')
        for k, v in model.state_dict().items():
            data[f'epoch_end{ep}/{k}'] = v.detach().clone().float().numpy()
        runner.epoch += 1
    data['lr'] = np.array(lrs, dtype=np.float64)
    data['momentum'] = np.array(moms, dtype=np.float64)
    data['loss'] = np.array(losses, dtype=np.float64)
    data['grad_norm'] = np.array([r['grad_norm'] for r in runner.log_buffer.rows], dtype=np.float64)
    data['param_names'] = np.array(names)
    data['cfg_json'] = np.array(json.dumps(CFG))
    data['accumulation'] = np.array(hooks[1].accumulation)
    data['ema_interval'] = np.array(hooks[0].interval)
    # schedule known answers at the recipe's real constants (yolov4l_coco_mosaic.py:124-139)
    its = [0, 1, 5000, 10000, 10001]
    kat = []
    for i in its:
        prog = i / 10000
        kat.append([prog + (1 - prog) * 10., prog + (1 - prog) * 0., prog + (1 - prog) * 0.95,
                    0.9999 * (1 - math.exp(-i / (10000 * 1)))])
    data['kat_iters'] = np.array(its)
    data['kat'] = np.array(kat, dtype=np.float64)
    data['kat_accum'] = np.array([[12, 1, math.ceil(64 / 12)], [8, 8, math.ceil(64 / 64)], [64, 8, math.ceil(64 / 512)]])
    for i, b in enumerate(runner.data_loader):
        data[f'batch{i}/img'] = b['img'].numpy()
        data[f'batch{i}/target'] = b['target'].numpy()
    out = os.path.join(HERE, 'hooks.npz')
    np.savez_compressed(out, **data)
    print('hooks', out, f'{os.path.getsize(out) / 1e3:.1f} kB', 'losses', [round(x, 4) for x in losses[:4]],
          'grad_norm', [round(float(x), 3) for x in data['grad_norm'][:3]])


if __name__ == '__main__':
    main()
