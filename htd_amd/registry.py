"""Registry / config surface of the drop-in: the reference builds every component with
`build_from_cfg(cfg, registry, default_args)` from dicts whose `type` names a registered class
(mmdet/models/builder.py:4-32, mmdet/core/bbox/builder.py, mmdet/core/anchor/builder.py; the
Registry itself is mmcv's).  Same names, same kwargs, same `type=` strings here, so
configs/htd/*.py load verbatim.
"""
import importlib.util
import os


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return key in self._module_dict

    def __repr__(self):
        return f'Registry(name={self._name}, items={sorted(self._module_dict)})'

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key, None)

    def _register_module(self, module_class, module_name=None, force=False):
        if not isinstance(module_class, type):
            raise TypeError(f'module must be a class, but got {type(module_class)}')
        name = module_name or module_class.__name__
        if not force and name in self._module_dict:
            raise KeyError(f'{name} is already registered in {self._name}')
        self._module_dict[name] = module_class

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._register_module(module, name, force)
            return module
        if isinstance(name, type):        # @REG.register_module without parentheses
            self._register_module(name)
            return name
        if not (name is None or isinstance(name, str)):
            raise TypeError(f'name must be a str, but got {type(name)}')

        def _register(cls):
            self._register_module(cls, name, force)
            return cls
        return _register


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
    if 'type' not in cfg:
        if default_args is None or 'type' not in default_args:
            raise KeyError(f'`cfg` or `default_args` must contain the key "type", but got {cfg}\n{default_args}')
    if not isinstance(registry, Registry):
        raise TypeError(f'registry must be a Registry object, but got {type(registry)}')
    args = dict(cfg)
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop('type')
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f'{obj_type} is not in the {registry.name} registry')
    elif isinstance(obj_type, type):
        obj_cls = obj_type
    else:
        raise TypeError(f'type must be a str or valid type, but got {type(obj_type)}')
    return obj_cls(**args)


class ConfigDict(dict):
    """Attribute-style dict; nested dicts/lists are wrapped recursively (train_cfg.rcnn is a list of
    dicts read as rcnn_train_cfg.assigner, htd_roi_head.py:106-108; .get() is used too, two_stage.py:151)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            return ConfigDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(ConfigDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, ConfigDict._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(f"'ConfigDict' object has no attribute '{k}'")

    def __setattr__(self, k, v):
        self[k] = v

    def copy(self):
        return ConfigDict({k: v for k, v in self.items()})

    def to_dict(self):
        def un(v):
            if isinstance(v, dict):
                return {k: un(x) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return type(v)(un(x) for x in v)
            return v
        return un(self)


def _merge(base, child):
    """mmcv Config merge: child keys override; dicts merge recursively unless `_delete_=True`."""
    out = dict(base)
    for k, v in child.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get('_delete_', False):
            out[k] = _merge(out[k], v)
        else:
            if isinstance(v, dict):
                v = {a: b for a, b in v.items() if a != '_delete_'}
            out[k] = v
    return out


class Config:
    """Loads an mmcv-style python config file, following `_base_` inheritance
    (configs/htd/htd_resnet50_1x.py:1-4)."""

    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, '_cfg_dict', ConfigDict(cfg_dict or {}))
        object.__setattr__(self, 'filename', filename)

    @staticmethod
    def _file2dict(filename):
        filename = os.path.abspath(os.path.expanduser(filename))
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        spec = importlib.util.spec_from_file_location('_htd_cfg', filename)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        cfg = {k: v for k, v in vars(mod).items() if not k.startswith('__') and not callable(v)
               and not isinstance(v, type(os))}
        bases = cfg.pop('_base_', [])
        bases = [bases] if isinstance(bases, str) else bases
        merged = {}
        for b in bases:
            merged = _merge(merged, Config._file2dict(os.path.join(os.path.dirname(filename), b)))
        return _merge(merged, cfg)

    @staticmethod
    def fromfile(filename):
        return Config(Config._file2dict(filename), filename)

    def merge_from_dict(self, options):
        """--cfg-options k.a.b=v (tools/train.py:55-60,86-87)."""
        nested = {}
        for full, v in options.items():
            d = nested
            keys = full.split('.')
            for k in keys[:-1]:
                d = d.setdefault(k, {})
            d[keys[-1]] = v
        object.__setattr__(self, '_cfg_dict', ConfigDict(_merge(self._cfg_dict.to_dict(), nested)))

    def __getattr__(self, k):
        return getattr(self._cfg_dict, k)

    def __getitem__(self, k):
        return self._cfg_dict[k]

    def __setattr__(self, k, v):
        self._cfg_dict[k] = v

    def __contains__(self, k):
        return k in self._cfg_dict

    def get(self, k, default=None):
        return self._cfg_dict.get(k, default)


BACKBONES = Registry('backbone')
NECKS = Registry('neck')
ROI_EXTRACTORS = Registry('roi_extractor')
SHARED_HEADS = Registry('shared_head')
HEADS = Registry('head')
LOSSES = Registry('loss')
DETECTORS = Registry('detector')
BBOX_ASSIGNERS = Registry('bbox_assigner')
BBOX_SAMPLERS = Registry('bbox_sampler')
BBOX_CODERS = Registry('bbox_coder')
ANCHOR_GENERATORS = Registry('Anchor generator')
IOU_CALCULATORS = Registry('IoU calculator')
CONV_LAYERS = Registry('conv layer')
ROI_LAYERS = Registry('roi layer')


def build(cfg, registry, default_args=None):
    if isinstance(cfg, list):
        import torch.nn as nn
        return nn.Sequential(*[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def build_backbone(cfg):
    return build(cfg, BACKBONES)


def build_neck(cfg):
    return build(cfg, NECKS)


def build_roi_extractor(cfg):
    return build(cfg, ROI_EXTRACTORS)


def build_head(cfg):
    return build(cfg, HEADS)


def build_loss(cfg):
    return build(cfg, LOSSES)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    return build(cfg, DETECTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_assigner(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_ASSIGNERS, default_args)


def build_sampler(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_SAMPLERS, default_args)


def build_bbox_coder(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_CODERS, default_args)


def build_anchor_generator(cfg, default_args=None):
    return build_from_cfg(cfg, ANCHOR_GENERATORS, default_args)


def build_iou_calculator(cfg, default_args=None):
    return build_from_cfg(cfg, IOU_CALCULATORS, default_args)
