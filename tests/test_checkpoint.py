"""Checkpoint round trips: file layouts the reference reads / writes, key revision, non-strict loading."""
import os

import pytest
import torch

import cmda_amd  # noqa: F401
from cmda_amd import checkpoint as ck
from cmda_amd.registry import build_backbone


def _b0():
    return build_backbone(dict(type='mit_b0', style='pytorch'))


def test_save_load_roundtrip_and_module_prefix(tmp_path):
    torch.manual_seed(0)
    a, b = _b0(), _b0()
    for p in a.parameters():
        torch.nn.init.normal_(p, std=0.1)
    f = os.path.join(tmp_path, 'iter_1.pth')
    saved = ck.save_checkpoint(a, f, meta=dict(iter=1))
    assert set(saved) == {'meta', 'state_dict'} and saved['meta']['iter'] == 1
    assert all(v.device.type == 'cpu' for v in saved['state_dict'].values())
    ck.load_checkpoint(b, f, strict=True)
    for (n, p), (_, q) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(p, q), n
    # DataParallel-style 'module.' prefixes are stripped (mmcv's default revise_keys)
    g = os.path.join(tmp_path, 'wrapped.pth')
    torch.save({'state_dict': {'module.' + k: v for k, v in a.state_dict().items()}}, g)
    c = _b0()
    ck.load_checkpoint(c, g, strict=True)
    assert all(torch.equal(p, q) for p, q in zip(a.state_dict().values(), c.state_dict().values()))


def test_pretrained_layouts_and_nonstrict(tmp_path):
    torch.manual_seed(1)
    a = _b0()
    for p in a.parameters():
        torch.nn.init.normal_(p, std=0.1)
    sd = dict(a.state_dict())
    sd['head.weight'] = torch.zeros(1000, 256)          # the ImageNet classifier head of the released MiT weights
    sd.pop('norm4.bias')
    for name, payload in (('bare.pth', sd), ('model.pth', {'model': sd}), ('sd.pth', {'state_dict': sd})):
        f = os.path.join(tmp_path, name)
        torch.save(payload, f)
        m = build_backbone(dict(type='mit_b0', style='pytorch', pretrained=f))
        m.init_weights()                                  # mix_transformer.py:343-357 path
        assert torch.equal(m.state_dict()['block1.0.attn.q.weight'], sd['block1.0.attn.q.weight'])
    missing, unexpected = ck.load_state_dict(_b0(), sd, strict=False)
    assert missing == ['norm4.bias'] and unexpected == ['head.weight']
    with pytest.raises(RuntimeError):
        ck.load_state_dict(_b0(), sd, strict=True)
    # parameters keep their storage (they may live inside a flat optimizer buffer)
    m = _b0()
    ptrs = [p.data_ptr() for p in m.parameters()]
    ck.load_state_dict(m, a.state_dict(), strict=True)
    assert ptrs == [p.data_ptr() for p in m.parameters()]
