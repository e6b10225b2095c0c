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


def test_load_refreshes_live_bf16_mirrors():
    """after FlatAdamW / attach_flat_store re-homed the parameters, runtime.w() returns the live bf16 mirror without
    consulting the cache -- a load must re-cast it from the new fp32 masters."""
    import cmda_amd.runtime as rt
    torch.manual_seed(2)
    a, b = _b0(), _b0()
    for p in a.parameters():
        torch.nn.init.normal_(p, std=0.1)
    for p in b.parameters():           # what FlatAdamW does on the GPU in bf16 mode: a persistent bf16 mirror per parameter
        p._cmda_bf16 = p.data.bfloat16()
    old = rt.compute_dtype()
    rt._state['dtype'] = torch.bfloat16
    try:
        ck.load_state_dict(b, a.state_dict(), strict=True)
        for (n, p), (_, q) in zip(b.named_parameters(), a.named_parameters()):
            assert torch.equal(rt.w(p), q.data.bfloat16()), n
    finally:
        rt._state['dtype'] = old


def test_flat_adamw_state_dict_roundtrip_and_release_strip(tmp_path):
    """optimizer state in torch.optim.AdamW's layout ({'state': {i: step/exp_avg/exp_avg_sq}, 'param_groups'}), restored into
    the flat buffers; save_checkpoint refuses optimizers without state_dict; function.py:28-37's stripping."""
    from cmda_amd import optim
    torch.manual_seed(3)
    m = _b0()
    opt = optim.FlatAdamW(m, lr=6e-5, weight_decay=0.01, custom_keys=dict(norm=dict(decay_mult=0.0)))
    opt.step_count = 7
    opt.flat_m.normal_()
    opt.flat_v.uniform_()
    f = os.path.join(tmp_path, 'iter_7.pth')
    ck.save_checkpoint(m, f, optimizer=opt, meta=dict(iter=7))
    saved = torch.load(f, weights_only=False)
    assert set(saved) == {'meta', 'state_dict', 'optimizer'}
    osd = saved['optimizer']
    names = [n for n, _ in m.named_parameters()]
    assert len(osd['param_groups']) == len(names) and [g['params'] for g in osd['param_groups']] == [[i] for i in range(len(names))]
    assert set(osd['param_groups'][0]) == {'lr', 'betas', 'eps', 'weight_decay', 'amsgrad', 'params'}
    wd = {n: g['weight_decay'] for n, g in zip(names, osd['param_groups'])}
    assert wd['block1.0.norm1.weight'] == 0.0 and wd['block1.0.attn.q.weight'] == 0.01
    i = names.index('block2.0.mlp.fc1.weight')
    assert osd['state'][i]['step'] == 7 and osd['state'][i]['exp_avg'].shape == dict(m.named_parameters())[names[i]].shape
    m2 = _b0()
    o2 = optim.FlatAdamW(m2, lr=6e-5, weight_decay=0.01, custom_keys=dict(norm=dict(decay_mult=0.0)))
    o2.load_state_dict(osd)
    assert o2.step_count == 7
    for pa, pb in zip(opt._order, o2._order):   # per parameter (the flat buffers also hold alignment padding)
        (la, ha), (lb, hb) = opt._slot(pa), o2._slot(pb)
        assert torch.equal(opt.flat_m[la:ha], o2.flat_m[lb:hb]) and torch.equal(opt.flat_v[la:ha], o2.flat_v[lb:hb])
    with pytest.raises(TypeError):
        ck.save_checkpoint(m, f, optimizer=object())
    full = {'model.backbone.a': 1, 'ema_model.backbone.a': 2, 'cyclegan_itrd2en.model.1.weight': 3, 'model.decode_head.b': 4}
    assert list(ck.strip_for_release(full)) == ['model.backbone.a', 'model.decode_head.b']


def test_refresh_frozen_rebuilds_channel_padded_copies(tgt):
    """the Motion-Extractor generator's first convolution is, in the bf16 mode, a FROZEN compute copy with its one
    input channel padded to 8 (runtime.wconv(ci_pad=8)).  A checkpoint load after a forward pass goes through
    runtime.refresh_frozen, which must rebuild that copy with the padding-aware batched re-layout -- the plain permute kernel
    indexed the [Co,1,7,7] master as [Co,8,7,7] (out-of-bounds read, garbage weights)."""
    import cmda_amd.runtime as rt
    from cmda_amd import _lib, ops
    rt.set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(2)
        p = torch.nn.Parameter(tgt.to(torch.randn(16, 1, 7, 7)))
        p._cmda_frozen = True
        q = torch.nn.Parameter(tgt.to(torch.randn(8, 4, 3, 3)))   # an unpadded frozen copy next to it
        q._cmda_frozen = True

        def want(w, cp):
            ref = torch.zeros(w.shape[0], w.shape[2], w.shape[3], cp)
            ref[..., :w.shape[1]] = w.detach().cpu().permute(0, 2, 3, 1)
            return ref.reshape(w.shape[0], -1).bfloat16().float()
        c1, c2 = rt.wconv(p, ci_pad=8), rt.wconv(q)
        assert torch.equal(c1.float().cpu(), want(p, 8)) and torch.equal(c2.float().cpu(), want(q, 4))
        with torch.no_grad():   # "checkpoint load": the masters change in place
            p.mul_(-3.0)
            q.add_(1.0)
        rt.refresh_frozen()
        assert rt.wconv(p, ci_pad=8) is c1 and rt.wconv(q) is c2, 'the copies keep their storage'
        assert torch.equal(c1.float().cpu(), want(p, 8)), 'padded frozen copy after refresh_frozen'
        assert torch.equal(c2.float().cpu(), want(q, 4))
        # the plain permute entry point refuses the padding bits instead of reading out of bounds
        with pytest.raises(_lib.CmdaError):
            ops.permute4(p.data, torch.empty_like(c1), (16, 8, 7, 7), (0, 2, 3, 1), flipmask=(2 << 8) | (1 << 16))
    finally:
        rt.set_compute_dtype(torch.float32)

