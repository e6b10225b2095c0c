"""Import shim that lets the reference's own, unmodified modules under /root/reference be imported in the
authoring container (where mmcv-full 1.3.7, timm 0.3.2, kornia, torchvision, h5py ... are not installed).

Used ONLY by tests/golden/make_golden.py to generate golden vectors.  Nothing here travels as a dependency of the
tests: the GPU box has no /root/reference, it only sees the committed .npz fixtures.

The few third-party names whose *semantics* matter on the hot path are restated from their documented behaviour
(mmcv ConvModule / DepthwiseSeparableConvModule / Registry / BaseModule, timm DropPath / trunc_normal_ / to_2tuple);
every other missing import becomes an inert stub.  See SURVEY.md section 8(c).
"""
import importlib
import importlib.abc
import importlib.machinery
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF = '/root/reference'
sys.dont_write_bytecode = True


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]  # behaves as an identity decorator
        return _Dummy()

    def __getattr__(self, n):
        if n.startswith('__'):
            raise AttributeError(n)
        return _Dummy()

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Dummy()


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = ('mmcv', 'timm', 'kornia', 'torchvision', 'h5py', 'hdf5plugin', 'prettytable', 'cityscapesscripts', 'cv2',
             'terminaltables', 'seaborn')

    def find_spec(self, fullname, path, target=None):
        if fullname.split('.')[0] in self.ROOTS and fullname not in sys.modules:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


# ---------------------------------------------------------------- restated third-party pieces
class Registry:
    def __init__(self, name, parent=None, **_):
        self.name = name
        self._module_dict = parent._module_dict if parent is not None else {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self._module_dict[name or cls.__name__] = cls
            return cls
        return deco(module) if module is not None else deco

    def get(self, key):
        return self._module_dict.get(key)

    def build(self, cfg, default_args=None):
        args = dict(cfg)
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        cls = self._module_dict[args.pop('type')]
        return cls(**args)


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg

    def init_weights(self):
        for m in self.children():
            if hasattr(m, 'init_weights'):
                m.init_weights()


def _ident_deco(*a, **k):
    def deco(fn):
        return fn
    return deco


class ConvModule(nn.Module):
    """mmcv 1.3.7 ConvModule restated: conv -> norm -> act, bias='auto'."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias='auto',
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True, **_):
        super().__init__()
        with_norm = norm_cfg is not None
        if bias == 'auto':
            bias = not with_norm
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        self.with_norm = with_norm
        if with_norm:
            assert norm_cfg['type'] in ('BN', 'SyncBN')
            self.bn = nn.BatchNorm2d(out_channels)
        self.with_activation = act_cfg is not None
        if self.with_activation:
            assert act_cfg['type'] == 'ReLU'
            self.activate = nn.ReLU(inplace=inplace)
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode='fan_out', nonlinearity='relu')
        if self.conv.bias is not None:
            nn.init.constant_(self.conv.bias, 0)

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            x = self.bn(x)
        if self.with_activation:
            x = self.activate(x)
        return x


class DepthwiseSeparableConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, norm_cfg=None,
                 act_cfg=dict(type='ReLU'), **_):
        super().__init__()
        self.depthwise_conv = ConvModule(in_channels, in_channels, kernel_size, stride=stride, padding=padding,
                                         dilation=dilation, groups=in_channels, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.pointwise_conv = ConvModule(in_channels, out_channels, 1, norm_cfg=norm_cfg, act_cfg=act_cfg)

    def forward(self, x):
        return self.pointwise_conv(self.depthwise_conv(x))


class DropPath(nn.Module):
    """timm 0.3.2 DropPath restated."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        r = keep + torch.rand(shape, dtype=x.dtype, device=x.device)
        r.floor_()
        return x.div(keep) * r


def install():
    if getattr(install, 'done', False):
        return
    install.done = True
    sys.meta_path.insert(0, _StubFinder())
    torch.Tensor.cuda = lambda self, *a, **k: self  # the head hard-codes .cuda() (decode_head.py:451,482,484)

    import mmcv  # noqa: F401  (stub)
    import mmcv.cnn
    import mmcv.cnn.bricks.registry
    import mmcv.parallel
    import mmcv.runner
    import mmcv.utils
    import timm.models.layers

    mmcv.utils.Registry = Registry
    mmcv.cnn.MODELS = Registry('model')
    mmcv.cnn.bricks.registry.ATTENTION = Registry('attention')
    mmcv.cnn.ConvModule = ConvModule
    mmcv.cnn.DepthwiseSeparableConvModule = DepthwiseSeparableConvModule
    mmcv.runner.BaseModule = BaseModule
    mmcv.runner.auto_fp16 = _ident_deco
    mmcv.runner.force_fp32 = _ident_deco
    mmcv.parallel.MMDistributedDataParallel = type('MMDistributedDataParallel', (), {})
    timm.models.layers.DropPath = DropPath
    timm.models.layers.to_2tuple = lambda v: v if isinstance(v, (tuple, list)) else (v, v)
    timm.models.layers.trunc_normal_ = nn.init.trunc_normal_

    # synthetic `mmseg` package tree rooted at the reference, without running mmseg/__init__.py (it asserts mmcv's version)
    for name in ('mmseg', 'mmseg.models', 'mmseg.models.backbones', 'mmseg.models.decode_heads', 'mmseg.models.losses',
                 'mmseg.models.segmentors', 'mmseg.models.fusion', 'mmseg.models.uda', 'mmseg.models.utils',
                 'mmseg.models.cyclegan', 'mmseg.ops', 'mmseg.core', 'mmseg.utils', 'mmseg.datasets'):
        m = types.ModuleType(name)
        m.__path__ = [REF + '/' + name.replace('.', '/')]
        sys.modules[name] = m
        if '.' in name:
            setattr(sys.modules[name.rsplit('.', 1)[0]], name.rsplit('.', 1)[1], m)
    import logging
    sys.modules['mmseg.utils'].get_root_logger = lambda *a, **k: logging.getLogger('mmseg')
    core = sys.modules['mmseg.core']
    core.add_prefix = lambda d, p: {f'{p}.{k}': v for k, v in d.items()}
    core.build_pixel_sampler = lambda *a, **k: None
    core.eval_metrics = None


def load(modname):
    """Import a reference module by dotted name, e.g. 'mmseg.models.backbones.mix_transformer'."""
    install()
    return importlib.import_module(modname)


def load_hotpath():
    """Load the hot-path modules in dependency order and return them in a namespace."""
    install()
    ns = types.SimpleNamespace()
    ns.builder = load('mmseg.models.builder')
    sys.modules['mmseg.models'].builder = ns.builder
    for k in ('BACKBONES', 'HEADS', 'LOSSES', 'SEGMENTORS', 'UDA', 'FUSION', 'build_segmentor', 'build_loss'):
        setattr(sys.modules['mmseg.models'], k, getattr(ns.builder, k))
    ns.wrappers = load('mmseg.ops.wrappers')
    sys.modules['mmseg.ops'].resize = ns.wrappers.resize
    ns.loss_utils = load('mmseg.models.losses.utils')
    ns.accuracy = load('mmseg.models.losses.accuracy')
    ns.ce = load('mmseg.models.losses.cross_entropy_loss')
    sys.modules['mmseg.models.losses'].accuracy = ns.accuracy.accuracy
    ns.mit = load('mmseg.models.backbones.mix_transformer')
    isa = types.ModuleType('mmseg.models.decode_heads.isa_head')
    isa.ISALayer = type('ISALayer', (nn.Module,), {})
    sys.modules['mmseg.models.decode_heads.isa_head'] = isa
    ns.decode_head = load('mmseg.models.decode_heads.decode_head')
    ns.aspp = load('mmseg.models.decode_heads.aspp_head')
    ns.segformer_head = load('mmseg.models.decode_heads.segformer_head')
    ns.sep_aspp = load('mmseg.models.decode_heads.sep_aspp_head')
    ns.daformer_head = load('mmseg.models.decode_heads.daformer_head')
    ns.avg_fusion = load('mmseg.models.fusion.attention_avg_fusion')
    ns.att_fusion = load('mmseg.models.fusion.attention_fusion')
    ns.seg_base = load('mmseg.models.segmentors.base')
    ns.encdec = load('mmseg.models.segmentors.encoder_decoder')
    ns.cyclegan = load('mmseg.models.cyclegan.cyclegan_model')
    ns.ds_utils = load('mmseg.datasets.utils')
    ns.dacs_transforms = load('mmseg.models.utils.dacs_transforms')
    return ns
