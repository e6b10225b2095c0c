"""mmcv-style Config loader (mmcv.Config is third-party and not installed; semantics restated from its documented
behaviour): python config files with `_base_` (str or list, relative paths), recursive dict merge, `_delete_=True`,
attribute access, and JSON child configs with `_base_` as written by the reference's launcher
(my_run_experiments.py:521-569).  configs/fusion/*.py of the reference load unchanged (tests/test_registry_config.py).
"""
import json
import os
import types

DELETE_KEY = '_delete_'
BASE_KEY = '_base_'


class ConfigDict(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value


def _to_configdict(obj):
    if isinstance(obj, dict):
        return ConfigDict({k: _to_configdict(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_configdict(v) for v in obj)
    return obj


def _merge(a, b):
    """merge child `a` into base `b` (returns a new dict)"""
    b = dict(b)
    for k, v in a.items():
        if isinstance(v, dict) and k in b and isinstance(b[k], dict) and not v.get(DELETE_KEY, False):
            b[k] = _merge(v, b[k])
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != DELETE_KEY}
            b[k] = v
    return b


def _load_file(path):
    path = os.path.abspath(path)
    if path.endswith('.json'):
        with open(path) as f:
            cfg = json.load(f)
    elif path.endswith('.py'):
        ns = {'__file__': path, '__name__': '_cmda_config_'}
        with open(path) as f:
            exec(compile(f.read(), path, 'exec'), ns)
        cfg = {k: v for k, v in ns.items()
               if not k.startswith('__') and not isinstance(v, (types.ModuleType, types.FunctionType, type))}
    else:
        raise IOError('Only py/json type are supported now!')
    if BASE_KEY in cfg:
        bases = cfg.pop(BASE_KEY)
        bases = bases if isinstance(bases, list) else [bases]
        base_cfg = {}
        for b in bases:
            sub = _load_file(os.path.join(os.path.dirname(path), b))
            dup = base_cfg.keys() & sub.keys()
            if dup:
                raise KeyError(f'Duplicate key is not allowed among bases: {sorted(dup)}')
            base_cfg.update(sub)
        cfg = _merge(cfg, base_cfg)
    return cfg


class Config(ConfigDict):
    @staticmethod
    def fromfile(filename):
        return Config(_to_configdict(_load_file(filename)))

    def merge_from_dict(self, options):
        """dotted keys, e.g. {'uda.sky_mask': None, 'model.decode_head.dropout_ratio': 0.0}"""
        for full, v in options.items():
            d = self
            keys = full.split('.')
            for k in keys[:-1]:
                d = d.setdefault(k, ConfigDict())
            d[keys[-1]] = _to_configdict(v)
        return self


def apply_launcher_defaults(cfg):
    """Keys the reference's launcher injects before tools/train.py sees the config (my_run_experiments.py:97-144,
    296-299) and that DACS.__init__ expects."""
    uda = cfg.setdefault('uda', ConfigDict())
    uda.setdefault('sky_mask', None)
    uda.setdefault('isr_another_fusion', False)
    model = cfg['model']
    fim = model.get('fusion_isr_module')
    if fim is not None and not uda.get('isr_another_fusion') and not uda.get('fuse_both_ice_and_e'):
        model['fusion_isr_module'] = ConfigDict(type='')
    return cfg
