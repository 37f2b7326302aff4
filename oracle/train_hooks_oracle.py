"""CPU restatement of the reference's optimizer-side training step.  TEST INFRASTRUCTURE ONLY:
imported by tests/ (and nothing in the product package); pinned against tests/golden/hooks.npz,
which was produced by the reference's own hooks (tests/golden/make_golden_hooks.py).

Every function is plain torch-CPU / python arithmetic and cites the reference lines it follows
(paths relative to /root/reference/mmdet/core/custom_hooks/).
"""
import math

import torch


def accumulation_steps(nominal_batch_size, samples_per_gpu, world_size):
    """accum_optim_hooks.py:29-34, ema_hooks.py:108-113."""
    return math.ceil(nominal_batch_size / (samples_per_gpu * world_size))


def warmup_value(cur_iter, warmup_iters, ratio, base):
    """warmup_hooks.py:42-58."""
    prog = cur_iter / warmup_iters
    return (prog + (1 - prog) * ratio) * base


def ema_momentum(momentum, cur_iter, warm_up, interval):
    """ema_hooks.py:90-91."""
    return momentum * (1 - math.exp(-cur_iter / (warm_up * interval)))


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (norm_type 2), as called through mmcv's
    OptimizerHook.clip_grads from accum_optim_hooks.py:47-49: returns (total_norm, grads scaled)."""
    total = torch.norm(torch.stack([torch.norm(g.detach(), 2.0) for g in grads]), 2.0)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, [g * coef for g in grads]


def sgd_nesterov(p, g, buf, lr, momentum, weight_decay, nesterov):
    """torch.optim.SGD single-tensor step, dampening 0 (the optimizer the recipe builds,
    configs/yolov4/yolov4l_coco_mosaic.py:108-115).  Returns (new p, new buf)."""
    if weight_decay != 0:
        g = g + weight_decay * p
    if momentum != 0:
        buf = g.clone() if buf is None else momentum * buf + g
        g = g + momentum * buf if nesterov else buf
    return p - lr * g, buf


class HookSimulator:
    """The three hooks acting on a model, in the recipe's priority order (EMA 'HIGH' first, then
    the optimizer hook, then the warm-up hook): state is held in plain dicts of CPU tensors.

    model: any nn.Module with ``train_step(data, optimizer) -> dict(loss=...)``; groups: list of
    dicts(lr, momentum, weight_decay, nesterov) one per parameter in named_parameters order."""

    def __init__(self, model, groups, cfg):
        self.model, self.cfg = model, cfg
        self.groups = [dict(g) for g in groups]
        self.names = [n for n, _ in model.named_parameters()]
        self.bufs = [None] * len(self.names)
        self.iter = 0
        self.lr_log, self.mom_log, self.norm_log = [], [], []
        # StateEMAHook.before_run (ema_hooks.py:52-64)
        self.ema = {k: v.detach().clone() for k, v in model.state_dict().items()}
        # DetailedLinearWarmUpHook.before_run (warmup_hooks.py:33-40)
        self.base_mom = [g['momentum'] for g in self.groups]
        self.base_lr_bias = {i: g['lr'] for i, (n, g) in enumerate(zip(self.names, self.groups)) if n.endswith('.bias')}
        self.base_lr_weight = {i: g['lr'] for i, (n, g) in enumerate(zip(self.names, self.groups))
                               if n.endswith('.weight')}
        self.accum = accumulation_steps(cfg['nominal_batch_size'], cfg['samples_per_gpu'], 1)
        self.interval = self.accum

    def swap(self):
        """ema_hooks.py:115-126."""
        sd = self.model.state_dict()
        with torch.no_grad():
            for k in list(self.ema.keys()):
                online = sd[k]
                tmp = online.detach().clone().float()
                online.copy_(self.ema[k].to(online.dtype))
                self.ema[k] = tmp

    def before_iter(self):
        c = self.cfg
        if self.iter <= c['warmup_iters']:
            for i, b in self.base_lr_bias.items():
                self.groups[i]['lr'] = warmup_value(self.iter, c['warmup_iters'], c['lr_bias_warmup_ratio'], b)
            for i, b in self.base_lr_weight.items():
                self.groups[i]['lr'] = warmup_value(self.iter, c['warmup_iters'], c['lr_weight_warmup_ratio'], b)
            for i, b in enumerate(self.base_mom):
                self.groups[i]['momentum'] = warmup_value(self.iter, c['warmup_iters'], c['momentum_warmup_ratio'], b)
        self.lr_log.append([g['lr'] for g in self.groups])
        self.mom_log.append([g['momentum'] for g in self.groups])

    def after_iter(self, loss):
        c = self.cfg
        params = [p for _, p in self.model.named_parameters()]
        # 1. StateEMAHook.after_train_iter (runs BEFORE the optimizer hook: priority HIGH) ema_hooks.py:80-98
        if (self.iter + 1) % self.interval == 0:
            sd = self.model.state_dict()
            m = ema_momentum(c['ema_momentum'], self.iter, c['ema_warm_up'], self.interval)
            for k in self.ema:
                if sd[k].dtype.is_floating_point:
                    self.ema[k] = self.ema[k].float() * m + sd[k].detach().float() * (1 - m)
                else:
                    self.ema[k] = sd[k].detach().float().to(self.ema[k].dtype)
        # 2. Fp16GradAccumulateOptimizerHook.after_train_iter accum_optim_hooks.py:37-60
        if self.iter % self.accum == 0:
            for p in params:
                p.grad = None
        loss.backward()
        if (self.iter + 1) % self.accum == 0:
            total, grads = clip_grad_norm([p.grad for p in params], c['max_norm'])
            self.norm_log.append(float(total))
            with torch.no_grad():
                for i, (p, g) in enumerate(zip(params, grads)):
                    h = self.groups[i]
                    newp, self.bufs[i] = sgd_nesterov(p.detach(), g, self.bufs[i], h['lr'], h['momentum'],
                                                      h['weight_decay'], h['nesterov'])
                    p.copy_(newp)
        self.iter += 1
