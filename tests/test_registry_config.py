"""The drop-in boundary: the reference's configs/fusion/*.py load unchanged and every `type` resolves to a cmda_amd
class with the reference's state_dict keys.  Reads /root/reference (authoring container only; skipped on the GPU box)."""
import json
import os

import pytest
import torch

import cmda_amd  # noqa: F401
from cmda_amd.config import Config, apply_launcher_defaults
from cmda_amd.registry import MODELS, build_train_model

REF = '/root/reference/configs/fusion'
HERE = os.path.dirname(os.path.abspath(__file__))


def test_registry_keys():
    for key in ('mit_b5', 'MixVisionTransformer', 'FusionEncoderDecoder', 'EncoderDecoder', 'AttentionAvgFusion',
                'AttentionFusion', 'DAFormerHeadFusion', 'DAFormerHead', 'CrossEntropyLoss', 'DACS'):
        assert key in MODELS, key


def test_config_merge_semantics(tmp_path):
    (tmp_path / 'base.py').write_text("model = dict(a=1, head=dict(type='x', k=3, fusion_cfg=dict(type='conv', kernel_size=1)))\nlr = 0.1\n")
    (tmp_path / 'child.py').write_text("_base_ = ['base.py']\nmodel = dict(head=dict(fusion_cfg=dict(_delete_=True, type='aspp', sep=True)))\nname = '{}_x'.format('ab')\n")
    cfg = Config.fromfile(str(tmp_path / 'child.py'))
    assert cfg.model.a == 1 and cfg.model.head.k == 3 and cfg.lr == 0.1 and cfg.name == 'ab_x'
    assert cfg.model.head.fusion_cfg == dict(type='aspp', sep=True)
    (tmp_path / 'gen.json').write_text(json.dumps({'_base_': ['child.py'], 'uda': {'sky_mask': None}, 'lr': 0.2}))
    cfg = Config.fromfile(str(tmp_path / 'gen.json'))
    assert cfg.lr == 0.2 and cfg.uda.sky_mask is None and cfg.model.head.fusion_cfg.sep is True


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference configs are only present in the authoring container')
@pytest.mark.parametrize('name', ['cs2dsec_image+events_together_b5.py', 'cs2dz_image+raw-isr_b5.py'])
def test_reference_fusion_configs_build(name):
    cfg = apply_launcher_defaults(Config.fromfile(os.path.join(REF, name)))
    assert cfg.model.type == 'FusionEncoderDecoder' and cfg.uda.type == 'DACS'
    cfg.model.pretrained = None  # pretrained/mit_b5.pth is an external download
    cfg.uda.cyclegan_itrd2en_path = 'random' if cfg.uda.get('cyclegan_itrd2en_path') else ''
    with torch.device('meta'):
        model = build_train_model(cfg)
    assert type(model).__name__ == 'DACS'
    keys = sorted(k[len('model.'):] for k in model.state_dict().keys() if k.startswith('model.'))
    if 'together' in name:
        with open(os.path.join(HERE, 'golden', 'segmentor_keys.json')) as f:
            ref_keys = json.load(f)
        assert [k for k in keys if not k.startswith('fusion_isr_module')] == ref_keys
    n = sum(p.numel() for p in model.model.parameters())
    assert n > 170e6
