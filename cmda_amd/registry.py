"""The drop-in boundary: one `MODELS` registry aliased as BACKBONES / NECKS / HEADS / LOSSES / SEGMENTORS / UDA /
FUSION plus the `build_*` helpers, with the semantics of the reference's mmseg/models/builder.py:10-79 (which sits on
mmcv.utils.Registry -- third-party, restated here: `build(cfg, default_args)` pops `type`, looks the class up and
calls `cls(**cfg)` with default_args applied via setdefault).  configs/fusion/* resolve against this registry
unchanged (tests/test_registry_config.py)."""
import warnings


class Registry:
    def __init__(self, name, parent=None):
        self.name = name
        self._module_dict = parent._module_dict if parent is not None else {}

    def __contains__(self, key):
        return key in self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def register_module(self, name=None, force=False, module=None):
        def _register(cls):
            key = name or cls.__name__
            if not force and key in self._module_dict and self._module_dict[key] is not cls:
                raise KeyError(f'{key} is already registered in {self.name}')
            self._module_dict[key] = cls
            return cls
        return _register(module) if module is not None else _register

    def build(self, cfg, default_args=None):
        if not isinstance(cfg, dict):
            raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
        if 'type' not in cfg and not (default_args and 'type' in default_args):
            raise KeyError('`cfg` or `default_args` must contain the key "type"')
        args = dict(cfg)
        if default_args is not None:
            for k, v in default_args.items():
                args.setdefault(k, v)
        obj_type = args.pop('type')
        if isinstance(obj_type, str):
            cls = self.get(obj_type)
            if cls is None:
                raise KeyError(f'{obj_type} is not in the {self.name} registry')
        else:
            cls = obj_type
        return cls(**args)


MODELS = Registry('models')
BACKBONES = NECKS = HEADS = LOSSES = SEGMENTORS = UDA = FUSION = MODELS
DATASETS = Registry('dataset')
PIPELINES = Registry('pipeline')


def build_from_cfg(cfg, registry, default_args=None):
    """mmcv.utils.build_from_cfg (third-party, restated): registry.build with default_args applied via setdefault"""
    return registry.build(cfg, default_args)


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_fusion(cfg):
    return FUSION.build(cfg)


def build_segmentor(cfg, train_cfg=None, test_cfg=None):
    if train_cfg is not None or test_cfg is not None:
        warnings.warn('train_cfg and test_cfg is deprecated, please specify them in model', UserWarning)
    assert cfg.get('train_cfg') is None or train_cfg is None, 'train_cfg specified in both outer field and model field '
    assert cfg.get('test_cfg') is None or test_cfg is None, 'test_cfg specified in both outer field and model field '
    return SEGMENTORS.build(cfg, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_train_model(cfg, train_cfg=None, test_cfg=None):
    """builder.py:47-65: wrap cfg.model into cfg.uda and build the UDA class when a `uda` section exists."""
    if 'uda' in cfg:
        cfg['uda']['model'] = cfg['model']
        cfg['uda']['max_iters'] = cfg['runner']['max_iters']
        return UDA.build(cfg['uda'], default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))
    return SEGMENTORS.build(cfg['model'], default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))
