"""Minimal registry / config surface the YOLOv4 path is plugged into.

The reference resolves every ``type='...'`` string through mmcv registries
(``mmdet/models/builder.py:6-14``, ``mmdet/core/anchor/builder.py:3``,
``mmdet/core/bbox/builder.py:3-5``; mmcv ``Registry`` / ``build_from_cfg`` /
``Config`` are third-party and absent on the GPU box).  This module provides the
same call surface -- ``Registry.register_module(name=None, force=False,
module=None)``, ``Registry.build(cfg)``, ``build_from_cfg(cfg, registry,
default_args)``, ``ConfigDict`` attribute access, ``Config.fromfile`` for
python-file configs with ``_base_`` inheritance and ``_delete_`` -- so the
``model = dict(type='SingleStageDetector', backbone=dict(type='DarknetCSP', ...))``
blocks of ``configs/yolov4/*`` build unchanged.
"""
import copy
import inspect
import os


class ConfigDict(dict):
    """dict with attribute access (the head uses ``hasattr(train_cfg, ...)`` and
    ``cfg.score_thr``, yolocsp_head.py:124-142,374-376)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, ConfigDict):
            return v
        if isinstance(v, dict):
            return ConfigDict(v)
        if isinstance(v, list):
            return [ConfigDict._wrap(x) for x in v]
        if isinstance(v, tuple):
            return tuple(ConfigDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, ConfigDict._wrap(v))

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        try:
            del self[name]
        except KeyError:
            raise AttributeError(name)

    def copy(self):
        return ConfigDict(super().copy())

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def to_dict(self):
        def un(v):
            if isinstance(v, dict):
                return {k: un(x) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return type(v)(un(x) for x in v)
            return v
        return un(self)


def build_from_cfg(cfg, registry, default_args=None):
    """Instantiate ``cfg['type']`` from ``registry`` with the remaining keys as kwargs."""
    if not isinstance(cfg, dict):
        raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
    if 'type' not in cfg:
        if default_args is None or 'type' not in default_args:
            raise KeyError(f'`cfg` or `default_args` must contain the key "type", but got {cfg}')
    if not isinstance(registry, Registry):
        raise TypeError(f'registry must be a Registry, but got {type(registry)}')
    args = dict(cfg)
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop('type')
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f'{obj_type} is not in the {registry.name} registry')
    elif inspect.isclass(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError(f'type must be a str or valid type, but got {type(obj_type)}')
    try:
        return obj_cls(**args)
    except Exception as e:
        raise type(e)(f'{obj_cls.__name__}: {e}')


class Registry:
    """Name -> class table with mmcv's decorator / function registration forms."""

    def __init__(self, name, build_func=None, parent=None):
        self._name = name
        self._module_dict = {}
        self._children = []
        self.parent = parent
        self.build_func = build_func or (parent.build_func if parent is not None else build_from_cfg)
        if parent is not None:
            parent._children.append(self)

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return self.get(key) is not None

    def __repr__(self):
        return f'Registry(name={self._name}, items={sorted(self._module_dict)})'

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        if key in self._module_dict:
            return self._module_dict[key]
        if self.parent is not None:
            return self.parent.get(key)
        return None

    def build(self, *args, **kwargs):
        return self.build_func(*args, **kwargs, registry=self)

    def _register(self, cls, name=None, force=False):
        if not inspect.isclass(cls):
            raise TypeError(f'module must be a class, but got {type(cls)}')
        names = [cls.__name__] if name is None else ([name] if isinstance(name, str) else list(name))
        for n in names:
            if not force and n in self._module_dict:
                raise KeyError(f'{n} is already registered in {self._name}')
            self._module_dict[n] = cls

    def register_module(self, name=None, force=False, module=None):
        if not isinstance(force, bool):
            raise TypeError(f'force must be a boolean, but got {type(force)}')
        if module is not None:  # x.register_module(module=SomeClass)
            self._register(module, name, force)
            return module

        def _decorator(cls):
            self._register(cls, name, force)
            return cls
        return _decorator


# The registries of the path (same objects the reference aliases to one another).
MODELS = Registry('models')
BACKBONES = MODELS
NECKS = MODELS
HEADS = MODELS
LOSSES = MODELS
DETECTORS = MODELS
ANCHOR_GENERATORS = Registry('Anchor generator')
BBOX_CODERS = Registry('bbox_coder')
BBOX_ASSIGNERS = Registry('bbox_assigner')
BBOX_SAMPLERS = Registry('bbox_sampler')
ACTIVATION_LAYERS = Registry('activation layer')
NORM_LAYERS = Registry('norm layer')
HOOKS = Registry('hook')


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    """``mmdet/models/builder.py:47-58``."""
    assert cfg.get('train_cfg') is None or train_cfg is None, \
        'train_cfg specified in both outer field and model field'
    assert cfg.get('test_cfg') is None or test_cfg is None, \
        'test_cfg specified in both outer field and model field'
    return DETECTORS.build(cfg, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_anchor_generator(cfg, default_args=None):
    return build_from_cfg(cfg, ANCHOR_GENERATORS, default_args)


def build_bbox_coder(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_CODERS, default_args)


def build_assigner(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_ASSIGNERS, default_args)


def build_sampler(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_SAMPLERS, default_args)


# ---- python-file configs ---------------------------------------------------------
def _merge(base, child):
    """mmcv Config._merge_a_into_b: child overrides base; ``_delete_=True`` replaces."""
    out = copy.deepcopy(base)
    for k, v in child.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get('_delete_', False):
            out[k] = _merge(out[k], v)
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != '_delete_'}
            out[k] = copy.deepcopy(v)
    return out


class Config:
    """``Config.fromfile(path)`` for python-file configs (``_base_`` / ``_delete_``)."""

    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, '_cfg_dict', ConfigDict(cfg_dict or {}))
        object.__setattr__(self, '_filename', filename)

    @staticmethod
    def _file2dict(filename):
        filename = os.path.abspath(os.path.expanduser(filename))
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        scope = {'__file__': filename}
        with open(filename) as f:
            exec(compile(f.read(), filename, 'exec'), scope)
        cfg = {k: v for k, v in scope.items()
               if not k.startswith('__') and not inspect.ismodule(v) and not inspect.isfunction(v)}
        if '_base_' in cfg:
            bases = cfg.pop('_base_')
            bases = bases if isinstance(bases, (list, tuple)) else [bases]
            merged = {}
            for b in bases:
                bd = Config._file2dict(os.path.join(os.path.dirname(filename), b))
                dup = set(merged) & set(bd)
                if dup:
                    raise KeyError(f'Duplicate key is not allowed among bases: {sorted(dup)}')
                merged.update(bd)
            cfg = _merge(merged, cfg)
        return cfg

    @staticmethod
    def fromfile(filename):
        return Config(Config._file2dict(filename), filename=filename)

    @property
    def filename(self):
        return self._filename

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setattr__(self, name, value):
        self._cfg_dict[name] = value

    def __contains__(self, name):
        return name in self._cfg_dict

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def merge_from_dict(self, options):
        """``--cfg-options a.b=1`` style overrides."""
        nested = {}
        for full, v in options.items():
            d = nested
            parts = full.split('.')
            for p in parts[:-1]:
                d = d.setdefault(p, {})
            d[parts[-1]] = v
        object.__setattr__(self, '_cfg_dict', ConfigDict(_merge(self._cfg_dict.to_dict(), nested)))
