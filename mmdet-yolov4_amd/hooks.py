"""The three training hooks of the YOLOv4 recipe, registered under the reference's names, plus
the small runner protocol they are driven through.

Reference surface mirrored here (``HOOKS.register_module``; constructor arguments, the stage
methods that do work, and the arithmetic of every schedule):
  * ``Fp16GradAccumulateOptimizerHook`` -- ``mmdet/core/custom_hooks/accum_optim_hooks.py:9-60``
    (torch >= 1.6 branch; base class ``mmcv.runner.Fp16OptimizerHook`` is third-party, its
    GradScaler behaviour is restated: ``loss_scale='dynamic'`` -> scale 2**16, x2 every 2000
    clean steps, x0.5 and skip on overflow; a float -> static scale)
  * ``StateEMAHook`` -- ``mmdet/core/custom_hooks/ema_hooks.py:8-126``
  * ``DetailedLinearWarmUpHook`` -- ``mmdet/core/custom_hooks/warmup_hooks.py:5-59``
Wired in configs as ``optimizer_config`` / ``custom_hooks``
(``configs/yolov4/yolov4l_coco_mosaic.py:117-147``).

What differs is where the work runs: un-scale + clip + SGD step + loss-scale update are four
kernel launches over the flat gradient arena with no host synchronisation, the EMA is one launch
over the whole state arena (``csrc/optim.hip``, ``flat_state.py``); the reference walks ~430
parameter groups and 658 state entries in Python for the same arithmetic.

The schedule arithmetic is exposed as pure functions (``warmup_factor``, ``ema_momentum``,
``accumulation_steps``) so it is testable without a device.
"""
import math

import os

import torch

from . import _lib
from ._lib import check
from .flat_state import FlatState
from .ops import stream_ptr
from .registry import HOOKS

# ---- pure schedule arithmetic ------------------------------------------------------------------


def warmup_factor(cur_iter, warmup_iters, ratio):
    """warmup_hooks.py:43-58: ``prog + (1 - prog) * ratio`` with ``prog = iter / warmup_iters``
    (multiplies the base lr / momentum while ``iter <= warmup_iters``)."""
    prog = cur_iter / warmup_iters
    return prog + (1 - prog) * ratio


def ema_momentum(momentum, cur_iter, warm_up, interval):
    """ema_hooks.py:90-91."""
    return momentum * (1 - math.exp(-cur_iter / (warm_up * interval)))


def accumulation_steps(nominal_batch_size, samples_per_gpu, world_size):
    """accum_optim_hooks.py:33-34 / ema_hooks.py:112-113."""
    return math.ceil(nominal_batch_size / (samples_per_gpu * world_size))


def _unwrap(model):
    return model.module if hasattr(model, 'module') and isinstance(model.module, torch.nn.Module) else model


def _world_size():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


# ---- hook base + priorities (mmcv.runner.Hook / Priority) -----------------------------------------
PRIORITY = dict(HIGHEST=0, VERY_HIGH=10, HIGH=30, ABOVE_NORMAL=40, NORMAL=50, BELOW_NORMAL=60, LOW=70,
                VERY_LOW=90, LOWEST=100)


class Hook:
    stages = ('before_run', 'before_train_epoch', 'before_train_iter', 'after_train_iter', 'after_train_epoch',
              'after_run')

    def before_run(self, runner):
        pass

    def after_run(self, runner):
        pass

    def before_epoch(self, runner):
        pass

    def after_epoch(self, runner):
        pass

    def before_iter(self, runner):
        pass

    def after_iter(self, runner):
        pass

    def before_train_epoch(self, runner):
        self.before_epoch(runner)

    def after_train_epoch(self, runner):
        self.after_epoch(runner)

    def before_train_iter(self, runner):
        self.before_iter(runner)

    def after_train_iter(self, runner):
        self.after_iter(runner)


# ---- a22 ---------------------------------------------------------------------------------------------
@HOOKS.register_module()
class Fp16GradAccumulateOptimizerHook(Hook):
    """Backward + (every ``accumulation`` iterations) un-scale, clip, step, loss-scale update.

    Gradients of the accumulated iterations are **summed**, not averaged (the reference's
    torch >= 1.6 branch never divides by ``accumulation``, accum_optim_hooks.py:37-60).
    The arena keeps the *scaled* gradients: un-scaling and clipping are folded into the one
    multiplier the step kernel applies (``ctrl[0]``), so ``p.grad`` is not rewritten in place."""

    def __init__(self, grad_clip=None, coalesce=True, bucket_size_mb=-1, loss_scale=512., distributed=True,
                 nominal_batch_size=None, accumulation=None, compute_dtype=None, grad_exchange=None):
        # compute_dtype: None leaves the model as it is (fp32 unless wrap_fp16_model was applied);
        # 'fp16' / 'bf16' (or the torch dtypes) make before_run wrap the model like the reference's
        # Fp16OptimizerHook.before_run does (mmcv wrap_fp16_model): 16-bit activations, fp32 master weights
        self.compute_dtype = {None: None, 'fp16': torch.float16, 'bf16': torch.bfloat16, 'fp32': torch.float32,
                              torch.float16: torch.float16, torch.bfloat16: torch.bfloat16,
                              torch.float32: torch.float32}[compute_dtype]
        # grad_exchange: None (YV4_GRAD_EXCHANGE or 'allreduce') | 'allreduce' | 'direct' | 'direct_bf16', see
        # dist.GradReducer
        self.grad_exchange = grad_exchange
        self.grad_clip = grad_clip
        self.coalesce = coalesce
        self.bucket_size_mb = bucket_size_mb
        self.distributed = distributed
        self.accumulation = 1
        self.nominal_batch_size = None
        if accumulation is not None:
            assert isinstance(accumulation, int) and accumulation > 0
            self.accumulation = accumulation
        elif nominal_batch_size is not None:
            self.accumulation = None
            self.nominal_batch_size = nominal_batch_size
        if grad_clip is not None:
            if grad_clip.get('norm_type', 2) != 2:
                raise NotImplementedError('grad_clip: only the L2 norm (norm_type=2) is implemented')
        # GradScaler(init_scale=2**16, growth_factor=2, backoff_factor=.5, growth_interval=2000)
        self.scaler_cfg = dict(init_scale=65536., growth_factor=2.0, backoff_factor=0.5, growth_interval=2000)
        self.dynamic = True
        if loss_scale == 'dynamic':
            pass
        elif isinstance(loss_scale, (int, float)):
            self.scaler_cfg['init_scale'] = float(loss_scale)
            self.dynamic = False
        elif isinstance(loss_scale, dict):
            # torch.cuda.amp.GradScaler's keys, or mmcv's legacy LossScaler spelling that the reference's own
            # yolov5_ddp configs use (configs/yolov5_ddp/*:21-23: init_scale, mode, scale_factor, scale_window)
            ls = dict(loss_scale)
            mode = ls.pop('mode', 'dynamic')
            if 'scale_factor' in ls:
                f = float(ls.pop('scale_factor'))
                ls.setdefault('growth_factor', f)
                ls.setdefault('backoff_factor', 1.0 / f)
            if 'scale_window' in ls:
                ls.setdefault('growth_interval', int(ls.pop('scale_window')))
            unknown = set(ls) - set(self.scaler_cfg)
            if unknown:
                raise TypeError(f'loss_scale: unexpected keys {sorted(unknown)}')
            self.scaler_cfg.update(ls)
            self.dynamic = mode == 'dynamic' 
        else:
            raise ValueError(f'loss_scale must be of type float, dict, or "dynamic", got {loss_scale}')
        self.scale_state = None
        self.ctrl = None
        self.reducer = None

    # device state is created lazily on the model's device
    def _ensure_state(self, runner):
        if self.scale_state is None:
            flat = FlatState.of(_unwrap(runner.model))
            self.flat = flat
            self.scale_state = torch.tensor([self.scaler_cfg['init_scale'], 0.], dtype=torch.float32,
                                            device=flat.device)
            self.ctrl = torch.zeros(4, dtype=torch.float32, device=flat.device)
            # 2 sums + one slot per workgroup of the deterministic mode's gradient norm (include/yv4.h)
            self.work = torch.zeros(2 + _lib.GRAD_PREPARE_MAX_WG, dtype=torch.float64, device=flat.device)
            if self.distributed and (_world_size() > 1 or os.environ.get('YV4_REDUCER_AT_WORLD1') == '1'):
                # (YV4_REDUCER_AT_WORLD1=1: build the reducer with one rank too -- tools/train_bench.py --overlap-report
                # measures where in backward each bucket becomes exchangeable; nothing is exchanged)
                from .dist import GradReducer
                mb = self.bucket_size_mb if self.bucket_size_mb and self.bucket_size_mb > 0 else 32
                self.reducer = GradReducer(flat, bucket_mb=mb, mode=self.grad_exchange)

    def before_run(self, runner):
        self._ensure_state(runner)
        if self.compute_dtype is not None:
            from .bricks import wrap_fp16_model
            wrap_fp16_model(_unwrap(runner.model), self.compute_dtype)
        meta = getattr(runner, 'meta', None)
        if meta and 'fp16' in meta and 'loss_scaler' in meta['fp16']:
            self.load_loss_scaler_state_dict(meta['fp16']['loss_scaler'])

    def before_train_epoch(self, runner):
        if self.accumulation is None:
            assert self.nominal_batch_size is not None
            samples_per_gpu = runner.data_loader.sampler.samples_per_gpu
            self.accumulation = accumulation_steps(self.nominal_batch_size, samples_per_gpu, _world_size())

    def loss_scale(self):
        """Current scale (synchronises)."""
        return float(self.scale_state[0].item())

    def after_train_iter(self, runner):
        self._ensure_state(runner)
        model = _unwrap(runner.model)
        first = runner.iter % self.accumulation == 0
        last = (runner.iter + 1) % self.accumulation == 0
        if first:
            # the reference zeroes through the model and through the optimizer; with both on the same gradient arena
            # one memset (and one walk over the parameters) does it
            fs = getattr(model, '_flat_state', None)
            if fs is not None and fs is getattr(runner.optimizer, 'flat', None):
                fs.zero_grad()
            else:
                model.zero_grad()
                runner.optimizer.zero_grad()
        if self.reducer is not None and last:
            self.reducer.arm()
        (runner.outputs['loss'] * self.scale_state[0]).backward()
        from .train_ops import join_side_streams
        join_side_streams()          # (the end-of-backward callback does it; it is dropped when a backward raises)
        if not last:
            return
        if self.reducer is not None:
            self.reducer.finish()
        max_norm = float(self.grad_clip['max_norm']) if self.grad_clip is not None else 0.
        L = _lib.lib()
        f = self.flat
        check(L.yv4_grad_prepare(f.grads.data_ptr(), f.n_param, self.scale_state.data_ptr(), max_norm,
                                 self.work.data_ptr(), self.ctrl.data_ptr(), stream_ptr()), 'yv4_grad_prepare')
        log = getattr(runner, 'log_buffer', None)
        if self.grad_clip is not None and log is not None:
            # the reference's float(grad_norm) drains the device here, between the backward pass and the update; the
            # copy is queued instead and waited for when the log buffer is read
            from .deferred import read_back_later
            log.update(read_back_later(self.ctrl, ['grad_norm', 'grad_scale'], lambda c: [c[1], 1.0 / c[3]]),
                       runner.outputs.get('num_samples', 1))
        runner.optimizer.step(ctrl=self.ctrl)
        if self.dynamic:
            check(L.yv4_loss_scale_update(self.scale_state.data_ptr(), self.ctrl.data_ptr(),
                                          float(self.scaler_cfg['growth_factor']),
                                          float(self.scaler_cfg['backoff_factor']),
                                          int(self.scaler_cfg['growth_interval']), stream_ptr()),
                  'yv4_loss_scale_update')

    # ---- checkpoint state (mmcv Fp16OptimizerHook keeps GradScaler.state_dict() in runner.meta['fp16']) -------
    def loss_scaler_state_dict(self):
        """``torch.cuda.amp.GradScaler.state_dict()`` layout: what the reference's base hook stores in
        ``runner.meta['fp16']['loss_scaler']`` and ``GradScaler.load_state_dict`` requires (one device read)."""
        if self.scale_state is None:
            scale, tracker = float(self.scaler_cfg['init_scale']), 0
        else:
            scale, tracker = self.scale_state.tolist()
        return dict(scale=float(scale), growth_factor=float(self.scaler_cfg['growth_factor']),
                    backoff_factor=float(self.scaler_cfg['backoff_factor']),
                    growth_interval=int(self.scaler_cfg['growth_interval']), _growth_tracker=int(tracker))

    def load_loss_scaler_state_dict(self, sd):
        for k in ('growth_factor', 'backoff_factor', 'growth_interval'):
            if k in sd:
                self.scaler_cfg[k] = sd[k]
        self.scale_state[0] = float(sd['scale'])
        self.scale_state[1] = float(sd.get('_growth_tracker', 0))

    def sync_meta(self, runner):
        """Write the plain loss-scaler dict into ``runner.meta``.  The reference does this after every iteration
        (a device read per step); here it happens when the state is wanted: before a checkpoint is written
        (``Runner.save_checkpoint``), at the end of every epoch and at the end of the run."""
        meta = getattr(runner, 'meta', None)
        if isinstance(meta, dict) and self.scale_state is not None:
            meta.setdefault('fp16', {})['loss_scaler'] = self.loss_scaler_state_dict()

    def after_train_epoch(self, runner):
        self.sync_meta(runner)

    def after_run(self, runner):
        self.sync_meta(runner)


@HOOKS.register_module()
class Fp16OptimizerHook(Fp16GradAccumulateOptimizerHook):
    """mmcv's ``Fp16OptimizerHook`` (third party; the base class of the fork's accumulate hook,
    accum_optim_hooks.py:9): what ``mmdet/apis/train.py:115-118`` builds from ``optimizer_config`` + the top-level
    ``fp16`` block of configs/yolov5_ddp/*.  One optimizer step per iteration = the accumulate hook with a window of 1."""

    def __init__(self, grad_clip=None, coalesce=True, bucket_size_mb=-1, loss_scale=512., distributed=True, **kw):
        super().__init__(grad_clip=grad_clip, coalesce=coalesce, bucket_size_mb=bucket_size_mb, loss_scale=loss_scale,
                         distributed=distributed, accumulation=1, **kw)


@HOOKS.register_module()
class OptimizerHook(Fp16GradAccumulateOptimizerHook):
    """mmcv's plain ``OptimizerHook(grad_clip)`` (``apis/train.py:119-120``): no loss scaling -- the same fused
    clip + step launches with a static scale of 1."""

    def __init__(self, grad_clip=None, **kw):
        kw.setdefault('distributed', True)
        super().__init__(grad_clip=grad_clip, loss_scale=1.0, accumulation=1, **kw)


def build_optimizer_hook(cfg, distributed=True):
    """``mmdet/apis/train.py:115-122``: the optimizer hook a config asks for.  ``cfg`` needs ``optimizer_config`` and
    optionally ``fp16`` (mapping or Config)."""
    get = cfg.get if hasattr(cfg, 'get') else (lambda k, d=None: getattr(cfg, k, d))
    oc = dict(get('optimizer_config') or {})
    fp16 = get('fp16', None)
    if fp16 is not None:
        return Fp16OptimizerHook(**oc, **dict(fp16), distributed=distributed)
    if 'type' not in oc:
        return OptimizerHook(**oc)
    from .registry import build_from_cfg
    return build_from_cfg(oc, HOOKS)


# ---- a23 ---------------------------------------------------------------------------------------------
@HOOKS.register_module()
class StateEMAHook(Hook):
    """EMA of every state-dict entry, kept as model buffers ``ema_<name with '.' -> '_'>`` so that
    checkpoints carry them (ema_hooks.py:52-64).  The buffers are views of one EMA arena laid out
    like the model's FlatState; integer entries (``num_batches_tracked``) are tracked as floats
    (what the reference's buffers become after the first swap, ema_hooks.py:118-126)."""

    def __init__(self, momentum=0.9999, interval=None, nominal_batch_size=None, warm_up=2000, resume_from=None):
        self.interval = 1
        self.nominal_batch_size = None
        self.warm_up = warm_up
        if interval is not None:
            assert isinstance(interval, int) and interval > 0
            self.interval = interval
        elif nominal_batch_size is not None:
            self.interval = None
            self.nominal_batch_size = nominal_batch_size
        assert momentum > 0 and momentum < 1
        self.momentum = momentum
        self.checkpoint = resume_from

    def before_run(self, runner):
        model = _unwrap(runner.model)
        flat = FlatState.of(model)
        self.flat = flat
        self.ema = flat.values.clone()
        self.ema_ints = flat.ints.to(torch.float32)
        from .flat_state import _view_as_param
        views = {}
        for seg in flat.param_segments + flat.buffer_segments:
            views[seg.name] = _view_as_param(self.ema, seg)
        for seg in flat.int_segments:
            views[seg.name] = self.ema_ints[seg.offset:seg.offset + seg.numel].view(seg.shape)
        # registered in state_dict order, like the reference (checkpoint key order)
        self.param_ema_mapping = {}
        for name in list(model.state_dict().keys()):
            if name not in views:
                raise RuntimeError(f'StateEMAHook: state entry {name} is not part of the FlatState')
            bname = f"ema_{name.replace('.', '_')}"
            self.param_ema_mapping[name] = bname
            model.register_buffer(bname, views[name])
        # ema_hooks.py:66-74: resume AFTER the ema_ buffers exist, from the hook's own argument or from the
        # ``resume_from`` of the config text the training script stored in ``runner.meta['config']``
        checkpoint = self.checkpoint
        if checkpoint is None:
            cfg_text = (getattr(runner, 'meta', None) or {}).get('config')
            if cfg_text is not None:
                cfg_dict = dict()
                exec(cfg_text, cfg_dict)
                checkpoint = cfg_dict.get('resume_from')
        if checkpoint is not None:
            runner.resume(checkpoint)

    def current_momentum(self, cur_iter):
        return ema_momentum(self.momentum, cur_iter, self.warm_up, self.interval)

    def after_train_iter(self, runner):
        if (runner.iter + 1) % self.interval != 0:
            return
        m = self.current_momentum(runner.iter)
        f = self.flat
        check(_lib.lib().yv4_ema_update(self.ema.data_ptr(), f.values.data_ptr(), f.n_state, float(m),
                                        stream_ptr()), 'yv4_ema_update')
        if f.n_int:
            self.ema_ints.copy_(f.ints)

    def after_train_epoch(self, runner):
        self._swap_ema_parameters(runner)

    def before_train_epoch(self, runner):
        if self.interval is None:
            assert self.nominal_batch_size is not None
            samples_per_gpu = runner.data_loader.sampler.samples_per_gpu
            self.interval = accumulation_steps(self.nominal_batch_size, samples_per_gpu, _world_size())
        self._swap_ema_parameters(runner)

    def _swap_ema_parameters(self, runner):
        f = self.flat
        with torch.no_grad():
            tmp = f.values.clone()
            f.values.copy_(self.ema)
            self.ema.copy_(tmp)
            if f.n_int:
                tmp_i = f.ints.to(torch.float32)
                f.ints.copy_(self.ema_ints.to(torch.int64))
                self.ema_ints.copy_(tmp_i)
        f.bump_versions()


# ---- a24 ---------------------------------------------------------------------------------------------
@HOOKS.register_module()
class DetailedLinearWarmUpHook(Hook):
    """Per-parameter linear warm-up: bias lr ``ratio_b -> 1``, weight lr ``ratio_w -> 1``,
    momentum ``ratio_m -> 1`` over ``warmup_iters`` iterations; needs one param group per
    parameter in ``named_parameters()`` order (warmup_hooks.py:24-40)."""

    def __init__(self, warmup_iters=10000, lr_weight_warmup_ratio=0., lr_bias_warmup_ratio=10.,
                 momentum_warmup_ratio=0.95):
        self.warmup_iters = warmup_iters
        self.lr_weight_warmup_ratio = lr_weight_warmup_ratio
        self.lr_bias_warmup_ratio = lr_bias_warmup_ratio
        self.momentum_warmup_ratio = momentum_warmup_ratio
        self.bias_base_lr = {}
        self.weight_base_lr = {}
        self.base_momentum = {}

    def before_run(self, runner):
        model = runner.model
        if len(runner.optimizer.param_groups) != len([*model.parameters()]):
            logger = getattr(runner, 'logger', None)
            if logger is not None:
                logger.warning('optimizer config does not support preheat because'
                               ' it is not using seperate param-group for each parameter')
            return
        for group_ind, (name, param) in enumerate(model.named_parameters()):
            group = runner.optimizer.param_groups[group_ind]
            self.base_momentum[group_ind] = group['momentum']
            if name.endswith('.bias'):
                self.bias_base_lr[group_ind] = group['lr']
            elif name.endswith('.weight'):
                self.weight_base_lr[group_ind] = group['lr']

    def before_train_iter(self, runner):
        if runner.iter <= self.warmup_iters:
            groups = runner.optimizer.param_groups
            fb = warmup_factor(runner.iter, self.warmup_iters, self.lr_bias_warmup_ratio)
            fw = warmup_factor(runner.iter, self.warmup_iters, self.lr_weight_warmup_ratio)
            fm = warmup_factor(runner.iter, self.warmup_iters, self.momentum_warmup_ratio)
            for group_ind, bias_base in self.bias_base_lr.items():
                groups[group_ind]['lr'] = fb * bias_base
            for group_ind, weight_base in self.weight_base_lr.items():
                groups[group_ind]['lr'] = fw * weight_base
            for group_ind, momentum_base in self.base_momentum.items():
                groups[group_ind]['momentum'] = fm * momentum_base


# ---- lr_config = dict(policy='CosineAnnealing', min_lr_ratio=0.2) ----------------------------------------
@HOOKS.register_module()
class CosineAnnealingLrUpdaterHook(Hook):
    """mmcv's epoch-based cosine annealing (third-party; restated): at the start of every epoch
    ``lr = min + 0.5 (base - min)(1 + cos(pi * epoch / max_epochs))`` per group from its
    ``initial_lr``.  No warm-up of its own: the recipe warms up through
    ``DetailedLinearWarmUpHook`` (``configs/yolov4/yolov4l_coco_mosaic.py:124-139``)."""

    def __init__(self, min_lr=None, min_lr_ratio=None, by_epoch=True, **kwargs):
        assert (min_lr is None) ^ (min_lr_ratio is None)
        self.min_lr, self.min_lr_ratio, self.by_epoch = min_lr, min_lr_ratio, by_epoch
        self.base_lr = []

    @staticmethod
    def annealing_cos(start, end, factor):
        return end + 0.5 * (start - end) * (math.cos(math.pi * factor) + 1)

    def before_run(self, runner):
        for g in runner.optimizer.param_groups:
            g.setdefault('initial_lr', g['lr'])
        self.base_lr = [g['initial_lr'] for g in runner.optimizer.param_groups]

    def _set(self, runner, progress, max_progress):
        for g, base in zip(runner.optimizer.param_groups, self.base_lr):
            target = base * self.min_lr_ratio if self.min_lr_ratio is not None else self.min_lr
            g['lr'] = self.annealing_cos(base, target, progress / max_progress)

    def before_train_epoch(self, runner):
        if self.by_epoch:
            self._set(runner, runner.epoch, runner.max_epochs)

    def before_train_iter(self, runner):
        if not self.by_epoch:
            self._set(runner, runner.iter, runner.max_iters)


# ---- the runner protocol the hooks are driven through ------------------------------------------------
class _Sampler:
    def __init__(self, samples_per_gpu):
        self.samples_per_gpu = samples_per_gpu


class BatchSource:
    """Anything iterable over ``train_step`` data dicts, plus the one attribute the hooks read
    from mmcv's loader: ``sampler.samples_per_gpu``."""

    def __init__(self, batches, samples_per_gpu):
        self.batches = batches
        self.sampler = _Sampler(samples_per_gpu)

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


class LogBuffer:
    def __init__(self):
        self.history = []

    def update(self, values, count=1):
        # a DeferredLogVars stays as it is (copying it would wait for the device)
        self.history.append((values if getattr(values, 'pending', False) else dict(values), count))


class Runner:
    """The slice of mmcv's ``EpochBasedRunner`` the three hooks touch: ``model``, ``optimizer``,
    ``iter``, ``epoch``, ``max_epochs``, ``outputs``, ``data_loader``, ``log_buffer``, ``meta``,
    hooks called in priority order at each stage, ``outputs = model.train_step(data, optimizer)``
    per iteration (mmcv ``epoch_based_runner.py`` ``run_iter``/``train``)."""

    def __init__(self, model, optimizer, logger=None, meta=None, max_epochs=1):
        self.model, self.optimizer, self.logger = model, optimizer, logger
        self.meta = meta if meta is not None else {}
        self.max_epochs = max_epochs
        self.iter = 0
        self.epoch = 0
        self.inner_iter = 0
        self.outputs = None
        self.data_loader = None
        self.log_buffer = LogBuffer()
        self._hooks = []

    def register_hook(self, hook, priority='NORMAL'):
        pr = PRIORITY[priority] if isinstance(priority, str) else int(priority)
        hook.priority = pr
        # mmcv inserts after the last hook of equal or higher priority
        i = len(self._hooks)
        while i > 0 and self._hooks[i - 1].priority > pr:
            i -= 1
        self._hooks.insert(i, hook)

    def register_hook_from_cfg(self, cfg):
        from .registry import build_from_cfg
        cfg = dict(cfg)
        priority = cfg.pop('priority', 'NORMAL')
        self.register_hook(build_from_cfg(cfg, HOOKS), priority)

    def call_hook(self, stage):
        for h in self._hooks:
            getattr(h, stage)(self)

    @property
    def max_iters(self):
        return self.max_epochs * len(self.data_loader)

    # ---- checkpoints: mmcv ``BaseRunner.save_checkpoint`` / ``resume`` (``epoch_based_runner.py``,
    # ``base_runner.py``): {'meta': {..., epoch, iter}, 'state_dict': weights on the CPU (the ``ema_*`` buffers
    # included: they are registered buffers, ema_hooks.py:52-64), 'optimizer': torch-layout state dict} ------------
    def save_checkpoint(self, out_dir, filename_tmpl='epoch_{}.pth', save_optimizer=True, meta=None,
                        create_symlink=True):
        import os
        import time
        from collections import OrderedDict
        for h in self._hooks:
            if hasattr(h, 'sync_meta'):
                h.sync_meta(self)
        meta = dict(meta or {})
        meta.update(self.meta or {})
        meta.update(epoch=self.epoch + 1, iter=self.iter, time=time.asctime())
        model = _unwrap(self.model)
        state = OrderedDict((k, v.detach().cpu()) for k, v in model.state_dict().items())
        ckpt = dict(meta=meta, state_dict=state)
        if save_optimizer and self.optimizer is not None:
            ckpt['optimizer'] = self.optimizer.state_dict()
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, filename_tmpl.format(self.epoch + 1))
        torch.save(ckpt, path)
        if create_symlink:
            link = os.path.join(out_dir, 'latest.pth')
            if os.path.islink(link) or os.path.exists(link):
                os.remove(link)
            os.symlink(os.path.basename(path), link)
        return path

    def load_checkpoint(self, filename, map_location='cpu', strict=False):
        ckpt = torch.load(filename, map_location=map_location, weights_only=False)
        state = ckpt['state_dict'] if 'state_dict' in ckpt else ckpt
        if all(k.startswith('module.') for k in state):
            state = {k[7:]: v for k, v in state.items()}
        model = _unwrap(self.model)
        missing, unexpected = model.load_state_dict(state, strict=strict)
        flat = getattr(model, '_flat_state', None)
        if flat is not None:
            flat.bump_versions()
        elif hasattr(model, 'invalidate_plans'):
            model.invalidate_plans()
        ckpt['_missing_keys'], ckpt['_unexpected_keys'] = list(missing), list(unexpected)
        return ckpt

    def resume(self, checkpoint, resume_optimizer=True, map_location='cpu'):
        ckpt = self.load_checkpoint(checkpoint, map_location=map_location)
        self.epoch = ckpt['meta']['epoch']
        self.iter = ckpt['meta']['iter']
        hook_msgs = dict((self.meta or {}).get('hook_msgs', {}))
        hook_msgs.update(ckpt['meta'].get('hook_msgs', {}))
        self.meta = dict(ckpt['meta'])                       # mmcv: "resume meta information meta"
        self.meta['hook_msgs'] = hook_msgs
        if 'optimizer' in ckpt and resume_optimizer and self.optimizer is not None:
            self.optimizer.load_state_dict(ckpt['optimizer'])
        return ckpt

    def run(self, data_loader, max_epochs=None):
        if max_epochs is not None:
            self.max_epochs = max_epochs
        self.data_loader = data_loader
        self.call_hook('before_run')
        while self.epoch < self.max_epochs:
            self.model.train()
            self.call_hook('before_train_epoch')
            for i, data in enumerate(data_loader):
                self.inner_iter = i
                self.call_hook('before_train_iter')
                self.outputs = self.model.train_step(data, self.optimizer)
                self.call_hook('after_train_iter')
                self.iter += 1
            self.call_hook('after_train_epoch')
            self.epoch += 1
        self.call_hook('after_run')
