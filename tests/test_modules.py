"""Module-level parity of the cmda_amd modules (HIP kernels through the C ABI) against the oracle and the golden
fixtures produced by the reference's own modules.  fp32 compute mode: tolerance 1e-3 relative (north star) -- in
practice ~1e-5; bf16 mode: documented looser bound."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
from weights import sample_grad, seeded_fill, seeded_randn  # noqa: E402

import cmda_amd.runtime as rt  # noqa: E402
from cmda_amd import backbones as bb  # noqa: E402
from conftest import assert_close, assert_close_fingerprint, assert_close_robust, check_ge, check_le  # noqa: E402


def gold(name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(HERE, 'golden', name + '.npz')).items()}


def check_grads(module, g, rtol, n=2048, atol=1e-6, outlier_frac=0.0, outlier_rtol=0.05):
    seen = 0
    for name, p in module.named_parameters():
        key = 'grad.' + name
        if key in g:
            assert p.grad is not None, name
            assert_close_fingerprint(sample_grad(p.grad, n), g[key], rtol, atol=atol, name=key, outlier_frac=outlier_frac, outlier_rtol=outlier_rtol)
            seen += 1
    assert seen == sum(k.startswith('grad.') for k in g)


@pytest.fixture
def mode(request, tgt):
    dt = getattr(request, 'param', torch.float32)
    rt.set_compute_dtype(dt)
    yield dt
    rt.set_compute_dtype(torch.float32)


BLOCK_CFGS = {'s1': (64, 1, 8, 16, 16), 's2': (128, 2, 4, 8, 16), 's3': (320, 5, 2, 8, 8), 's4': (512, 8, 1, 4, 4),
              'f1': (128, 1, 4, 8, 8)}


@pytest.mark.parametrize('mode', [torch.float32, torch.bfloat16], indirect=True)
@pytest.mark.parametrize('tag', list(BLOCK_CFGS))
def test_block_golden(tgt, mode, tag):
    from functools import partial
    dim, heads, sr, H, W = BLOCK_CFGS[tag]
    g = gold('block_' + tag)
    m = bb.Block(dim, heads, 4, True, drop_path=0.0, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), sr_ratio=sr)
    seeded_fill(m, 11).train().to(tgt.device)
    x = tgt.to(seeded_randn((2, H * W, dim), 11, 'x')).requires_grad_(True)
    y = m(x, H, W)
    y.backward(tgt.to(seeded_randn(y.shape, 11, 'dy')))
    f32 = mode == torch.float32
    # bf16 bounds = 2.5x the worst value of the audited GPU runs (gpurun r05modules: y 4.0e-2 of 0.17 -> 7.5e-3 relative, dx 5.7e-2
    # of 0.31, gradient samples within 1.2e-2 of their largest element)
    assert_close(y, g['y'], 1e-4 if f32 else 2e-2, name='y')
    assert_close(x.grad, g['dx'], 1e-4 if f32 else 2.5e-2, name='dx')
    check_grads(m, g, 2e-4 if f32 else 3e-2, atol=1e-6 if f32 else 1e-3)


@pytest.mark.parametrize('mode', [torch.float32], indirect=True)
def test_mit_small_train_golden(tgt, mode):
    g = gold('mit_small_train')
    m = bb.MixVisionTransformer(patch_size=4, embed_dims=[64, 128, 320, 512], num_heads=[1, 2, 5, 8], qkv_bias=True,
                                norm_layer=__import__('functools').partial(torch.nn.LayerNorm, eps=1e-6),
                                depths=[1, 1, 1, 1], sr_ratios=[8, 4, 2, 1], drop_path_rate=0.0)
    seeded_fill(m, 22).train().to(tgt.device)
    outs = m(tgt.to(seeded_randn((2, 3, 64, 96), 22, 'img')))
    sum((o * tgt.to(seeded_randn(o.shape, 22, f'dy{i}'))).sum() for i, o in enumerate(outs)).backward()
    for i, o in enumerate(outs):
        assert_close(o, g[f'out{i}'], 1e-4, name=f'out{i}')
    check_grads(m, g, 5e-4)


def feats_nlc(tgt, B, H, W, seed, tag):
    out = []
    for i, (c, s) in enumerate(zip([64, 128, 320, 512], [4, 8, 16, 32])):
        f = seeded_randn((B, c, H // s, W // s), seed, f'{tag}{i}')
        out.append((tgt.to(f.permute(0, 2, 3, 1).reshape(-1, c).contiguous().to(rt.compute_dtype())), H // s, W // s))
    return out


HEAD_KW = dict(in_channels=[64, 128, 320, 512], in_index=[0, 1, 2, 3], channels=256, dropout_ratio=0.0, num_classes=19,
               norm_cfg=dict(type='BN', requires_grad=True), align_corners=False,
               loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0))


def decoder_params(**extra):
    d = dict(embed_dims=256, embed_cfg=dict(type='mlp', act_cfg=None, norm_cfg=None),
             embed_neck_cfg=dict(type='mlp', act_cfg=None, norm_cfg=None),
             fusion_cfg=dict(type='aspp', sep=True, dilations=(1, 6, 12, 18), pool=False, act_cfg=dict(type='ReLU'),
                             norm_cfg=dict(type='BN', requires_grad=True)))
    d.update(extra)
    return d


@pytest.mark.parametrize('mode', [torch.float32, torch.bfloat16], indirect=True)
def test_head_train_golden(tgt, mode):
    from cmda_amd import decode_heads as dh
    g = gold('head_train')
    head = dh.DAFormerHead(**HEAD_KW, decoder_params=decoder_params())
    seeded_fill(head, 31).train().to(tgt.device)
    B, H, W = 2, 64, 96
    feats = feats_nlc(tgt, B, H, W, 31, 'f')
    losses, logits, saved = head.fwd_train(feats, B, tgt.to(g['gt']), tgt.to(g['weight']))
    dfs = head.bwd_train(saved, B, gscale=tgt.to(torch.tensor([1.7])))
    f32 = mode == torch.float32
    assert_close(logits.permute(0, 3, 1, 2), g['logits'], 1e-4 if f32 else 4e-2, name='logits')
    assert_close(losses['loss_seg'], g['loss_seg'], 1e-5 if f32 else 5e-3, name='loss')
    assert_close(losses['acc_seg'], g['acc_seg'], 1e-6 if f32 else 0.3, name='acc')
    for i in range(4):
        c = dfs[i].shape[1]
        ref = g[f'dfeat{i}'].permute(0, 2, 3, 1).reshape(-1, c)
        assert_close(dfs[i], ref, 3e-4 if f32 else 0.4, atol=1e-8 if f32 else 2e-6, name=f'dfeat{i}', outlier_frac=1e-2 if f32 else 0.0)  # bf16: 3 train-mode BNs amplify rounding (12.7 % of range measured)
    check_grads(head, g, 5e-4 if f32 else 0.1, atol=1e-6 if f32 else 2e-3, outlier_frac=1e-2 if f32 else 0.0)   # (bf16: worst sample 3.5e-2 of its tensor's largest element, gpurun r05modules)
    if f32:
        for k, v in head.state_dict().items():
            if 'running' in k:
                assert_close(v, g['bn.' + k], 1e-4, name=k)


@pytest.mark.parametrize('mode', [torch.float32], indirect=True)
@pytest.mark.parametrize('joint', [False, True], ids=['per_branch', 'joint'])
def test_head_fusion_train_golden(tgt, mode, joint):
    """joint: the shared decoder run ONCE over the four feature sets (grouped BatchNorm statistics, Dropout2d on the image block
    only) must reproduce the reference's four sequential branch passes -- logits, loss mix, input and parameter gradients and
    the BatchNorm running statistics after the four ordered updates."""
    from cmda_amd import decode_heads as dh
    g = gold('head_fusion_train')
    head = dh.DAFormerHeadFusion(**HEAD_KW, decoder_params=decoder_params(train_type='cs2dsec_image+events_together',
                                                                          share_decoder=True))
    import json
    with open(os.path.join(HERE, 'golden', 'head_fusion_keys.json')) as f:
        assert sorted(head.state_dict().keys()) == json.load(f)
    seeded_fill(head, 41).train().to(tgt.device)
    B, H, W = 1, 64, 64
    inputs = {k: feats_nlc(tgt, B, H, W, 41, k) for k in ('f_image', 'f_events', 'f_fusion', 'f_img_self_res')}
    cfg = dict(loss_weight={'image': 0.5, 'events': 0.5, 'fusion': 0.5, 'img_self_res': 0.25})
    if joint:
        names = ('image', 'fusion', 'events', 'isr')
        fkey = {'image': 'f_image', 'fusion': 'f_fusion', 'events': 'f_events', 'isr': 'f_img_self_res'}
        feats = [(torch.cat([inputs[fkey[n]][i][0] for n in names]).contiguous(), inputs['f_image'][i][1], inputs['f_image'][i][2])
                 for i in range(4)]
        losses, logits, saved = head.fwd_train_joint(feats, names, B, tgt.to(g['gt']), None, cfg)
        dJ = head.bwd_train_joint(saved, B)
        dfs = {fkey[n]: {i: dJ[i][gi * (dJ[i].shape[0] // 4):(gi + 1) * (dJ[i].shape[0] // 4)] for i in range(4)}
               for gi, n in enumerate(names)}
    else:
        losses, logits, saved = head.fwd_train(inputs, B, tgt.to(g['gt']), None, cfg)
        dfs = head.bwd_train(saved, B)
    assert_close(losses['loss_seg'], g['loss_seg'], 1e-5, name='loss')
    assert_close(losses['acc_seg'], g['acc_seg'], 1e-6, name='acc')
    for k, v in logits.items():
        assert_close(v.permute(0, 3, 1, 2), g[k], 1e-4, name=k)
    # For this seed some BN+ReLU pre-activations lie within fp32 round-off of zero: under a different summation order
    # (atomics) a mask differs from the reference's, and train-mode BN's backward (batch means of dy) spreads that over
    # the whole branch at the 1e-3 level.  Which masks flip depends on the atomics' order (box to box: 3.9 % of the gradient's range
    # on ONE element of one branch was the worst seen), so the gate is the 99.9th percentile with a loose hard bound on the single
    # worst element (the affected branch's gradient is off by ~2 % of its range in the bulk too: BN's batch means carry the flip to
    # every pixel), and at least one of the four branches tight.
    tight = 0
    for k, d in dfs.items():
        worst = 0.0
        for i in range(4):
            c = d[i].shape[1]
            ref = g[f'd{k}{i}'].permute(0, 2, 3, 1).reshape(-1, c)
            assert_close_robust(d[i], ref, 6e-2, 0.15, name=f'd{k}{i}')   # (2.1e-2 / 3.9e-2 measured on the flip-affected branch)
            worst = max(worst, (d[i].float().cpu() - ref).abs().max().item() / ref.abs().max().item())
        tight += worst < 3e-4
    check_ge('branches with every input gradient within 3e-4', tight, 2)   # (3 of 4 measured on the GPU and the emulator)
    check_grads(head, g, 5e-4, outlier_frac=1.0, atol=2e-6, outlier_rtol=0.08)   # (flipped BN + ReLU masks: 3.0e-2 of the tensor's largest element, two runs)
    for k, v in head.state_dict().items():
        if 'running' in k and ('bn.' + k) in g:
            assert_close(v, g['bn.' + k], 1e-4, name=k)


@pytest.mark.parametrize('mode', [torch.float32], indirect=True)
@pytest.mark.parametrize('name', ['avg', 'cat'])
def test_fusion_modules_golden(tgt, mode, name):
    from cmda_amd import fusion as fu
    g = gold('fusion_' + name)
    cls = fu.AttentionAvgFusion if name == 'avg' else fu.AttentionFusion
    m = seeded_fill(cls(drop_path_rate=0.0), 91).train().to(tgt.device)
    fi, fe = feats_nlc(tgt, 1, 64, 64, 91, 'i'), feats_nlc(tgt, 1, 64, 64, 91, 'e')
    outs, saved = m.fwd(fi, fe, 1)
    for i, (o, H, W) in enumerate(outs):
        assert_close(o, g[f'out{i}'].permute(0, 2, 3, 1).reshape(H * W, -1), 1e-4, name=f'out{i}')
    # backward consistency against the oracle (no golden gradients stored for these modules)
    from oracle import fusion as ofu
    ref = seeded_fill((ofu.AttentionAvgFusion if name == 'avg' else ofu.AttentionFusion)(drop_path_rate=0.0), 91).train()
    ri = [seeded_randn((1, c, 64 // s, 64 // s), 91, f'i{k}').requires_grad_(True) for k, (c, s) in enumerate(zip([64, 128, 320, 512], [4, 8, 16, 32]))]
    re = [seeded_randn((1, c, 64 // s, 64 // s), 91, f'e{k}').requires_grad_(True) for k, (c, s) in enumerate(zip([64, 128, 320, 512], [4, 8, 16, 32]))]
    routs = ref(ri, re)
    dys = [seeded_randn(o.shape, 92, f'dy{k}') for k, o in enumerate(routs)]
    sum((o * d).sum() for o, d in zip(routs, dys)).backward()
    dfused = [tgt.to(d.permute(0, 2, 3, 1).reshape(-1, d.shape[1]).contiguous()) for d in dys]
    di, de = m.bwd(saved, dfused, 1)
    for k in range(4):
        c = di[k].shape[1]
        assert_close(di[k], ri[k].grad.permute(0, 2, 3, 1).reshape(-1, c), 2e-4, name=f'di{k}')
        assert_close(de[k], re[k].grad.permute(0, 2, 3, 1).reshape(-1, c), 2e-4, name=f'de{k}')
    for (n1, p), (n2, q) in zip(m.named_parameters(), ref.named_parameters()):
        assert n1 == n2
        assert_close(p.grad, q.grad, 5e-4, atol=1e-6, name=n1)


SEG_CFG = dict(type='FusionEncoderDecoder', pretrained=None,
               backbone_image=dict(type='mit_b5', style='pytorch', in_chans=3, drop_path_rate=0.0),
               backbone_events=dict(type='mit_b5', style='pytorch', in_chans=3, drop_path_rate=0.0),
               fusion_module=dict(type='AttentionAvgFusion', drop_path_rate=0.0),
               decode_head=dict(type='DAFormerHeadFusion', **HEAD_KW,
                                decoder_params=decoder_params(train_type='cs2dsec_image+events_together', share_decoder=True)),
               train_type='cs2dsec_image+events_together', train_cfg=dict(), test_cfg=dict(mode='whole'))


@pytest.mark.slow
@pytest.mark.parametrize('mode', [torch.float32], indirect=True)
def test_fusion_segmentor_golden(tgt, mode):
    """Full student (2 x MiT-B5 + fusion + 4 decoder passes + loss) and teacher pass vs the reference's own outputs."""
    import json
    import cmda_amd  # noqa: F401
    from cmda_amd.registry import build_segmentor
    if tgt.kind == 'emu' and not os.environ.get('CMDA_SLOW'):
        pytest.skip('9 min in the CPU emulator (set CMDA_SLOW=1); always runs on the GPU')
    g = gold('segmentor_train')
    model = build_segmentor(SEG_CFG)
    with open(os.path.join(HERE, 'golden', 'segmentor_keys.json')) as f:
        assert sorted(model.state_dict().keys()) == json.load(f)
    seeded_fill(model, 101).train().to(tgt.device)
    inputs = {k: tgt.to(seeded_randn((1, 3, 64, 64), 101, k)) for k in ('image', 'events', 'img_self_res')}
    fcfg = dict(loss_weight={'image': 0.5, 'events': 0.5, 'fusion': 0.5, 'img_self_res': 0.25})
    losses, pred = model.forward_train(inputs, tgt.to(g['gt']), return_feat=True, cfg=fcfg)
    losses['decode.loss_seg'].backward()
    assert_close(losses['decode.loss_seg'], g['loss_seg'], 1e-4, name='loss')
    assert_close(losses['decode.acc_seg'], g['acc_seg'], 1e-3, name='acc')
    for k, v in pred.items():
        assert_close(v, g[k], 1e-3, name=k)  # north-star bound: logits within 1e-3 relative
    check_grads(model, g, 5e-2, n=96, atol=1e-5)
    gt = gold('segmentor_teacher')
    model.eval()
    out = model.encode_decode(inputs['image'], inputs['events'], output_features=True, test_cfg=fcfg)
    for k, v in gt.items():
        assert_close(out[k], v, 1e-3, name=k)


@pytest.mark.parametrize('mode', [torch.float32, torch.bfloat16], indirect=True)
def test_generator_golden(tgt, mode):
    import json
    from cmda_amd import cyclegan as cg
    g = gold('generator')
    G = cg.ResnetGenerator()
    with open(os.path.join(HERE, 'golden', 'generator_keys.json')) as f:
        assert sorted(G.state_dict().keys()) == json.load(f)
    seeded_fill(G, 81).eval().to(tgt.device)
    y = G(tgt.to(seeded_randn((2, 1, 32, 48), 81, 'x')))
    assert_close(y, g['y'], 1e-4 if mode == torch.float32 else 0.1, name='generator')  # bf16: 23 conv+InstanceNorm layers, random weights


@pytest.mark.parametrize('mode', [torch.float32, torch.bfloat16], indirect=True)
def test_generator_fused_instance_norm_statistics(tgt, mode):
    """every InstanceNorm of the generator taking its statistics from the epilogue of the convolution in front of it
    (cmda_gemm_params_t.colstats; 64 x 64 input: all 23 conv / norm pairs have whole 256-row tiles per sample) against the same
    model with the separate statistics pass, and against the oracle"""
    from cmda_amd import cyclegan as cg, ops
    from oracle import cyclegan as ocg
    G = seeded_fill(cg.ResnetGenerator(), 83).eval().to(tgt.device)
    x = seeded_randn((2, 1, 64, 64), 83, 'x')
    calls = []
    orig = ops.bn_train_fwd2

    def spy(*a, **k):
        calls.append(k.get('stats_ws') is not None)
        return orig(*a, **k)
    prev = ops.BN_FUSED_STATS
    ops.bn_train_fwd2 = spy
    try:
        ops.BN_FUSED_STATS = True
        y_fused = G(tgt.to(x)).float().cpu()
        n_fused = sum(calls)
        ops.BN_FUSED_STATS = False
        y_sep = G(tgt.to(x)).float().cpu()
    finally:
        ops.BN_FUSED_STATS, ops.bn_train_fwd2 = prev, orig
    assert n_fused == 23 and sum(calls) == 23, calls
    ref = seeded_fill(ocg.ResnetGenerator(), 83).eval()
    with torch.no_grad():
        want = ref(x)
    tol = 1e-4 if mode == torch.float32 else 0.1
    assert_close(y_fused, want, tol, name='generator, fused statistics vs oracle')
    assert_close(y_fused, y_sep, 4e-5 if mode == torch.float32 else 0.05, name='generator, fused vs separate statistics pass')   # (1.0e-5 measured)


@pytest.mark.gpu
def test_eval_path_440x640_matches_oracle():
    """SURVEY 8f (next row): whole-image inference at the DSEC evaluation size 440x640 -- token grids that are not powers
    of two and Nk = 260 keys (beyond the fused-attention limit, so the GEMM + softmax path runs) -- against the oracle,
    then the mIoU bookkeeping over the two label maps."""
    from cmda_amd import metrics
    from cmda_amd.registry import build_segmentor
    from oracle import head as ohd, mit as omit, segmentor as oseg
    torch.manual_seed(0)
    dev = torch.device('cuda:0')
    depths = [1, 1, 1, 1]
    cfg = dict(type='EncoderDecoder',
               backbone=dict(type='MixVisionTransformer', embed_dims=[64, 128, 320, 512], num_heads=[1, 2, 5, 8],
                             qkv_bias=True, depths=depths, sr_ratios=[8, 4, 2, 1], drop_path_rate=0.0),
               decode_head=dict(type='DAFormerHead', in_channels=[64, 128, 320, 512], in_index=[0, 1, 2, 3], channels=256,
                                dropout_ratio=0.0, num_classes=19, norm_cfg=dict(type='BN'), align_corners=False,
                                decoder_params=dict(embed_dims=256, embed_cfg=dict(type='mlp'), embed_neck_cfg=dict(type='mlp'),
                                                    fusion_cfg=dict(type='aspp', sep=True, dilations=(1, 6, 12, 18), pool=False))))
    model = build_segmentor(cfg)
    ref = oseg.EncoderDecoder(omit.MixVisionTransformer(depths=depths, drop_path_rate=0.0, eps=1e-5), ohd.DAFormerHead(dropout_ratio=0.0))
    ref.load_state_dict(model.state_dict())
    model.to(dev).eval()
    ref.eval()
    img = torch.randn(1, 3, 440, 640)
    with torch.no_grad():
        want = ref.encode_decode(img)
    for dt, tol, agree_min in ((torch.float32, 1e-3, 0.9995), (torch.bfloat16, 6e-2, 0.97)):
        rt.set_compute_dtype(dt)
        try:
            got = model.encode_decode(img.to(dev)).float().cpu()
            pred = model.simple_test(img.to(dev))[0]
        finally:
            rt.set_compute_dtype(torch.float32)
        assert got.shape == want.shape == (1, 19, 440, 640)
        err = (got - want).abs().max().item() / want.abs().max().item()
        check_le(f'{dt}: logits relative error', err, tol, strict=True)
        agree = float((torch.from_numpy(pred) == want.argmax(1)[0]).float().mean())
        check_ge(f'{dt}: argmax agreement', agree, agree_min)
        r = metrics.mean_iou([torch.from_numpy(pred).to(dev)], [want.argmax(1)[0].to(dev)], 19, 255)
        assert r['aAcc'].item() >= agree_min and r['IoU'].device.type == 'cuda'


def test_patch_embed_channel_padding(tgt):
    """bf16 mode: the encoder's 3-channel input is laid out with 8 channels per pixel (zeros) and the 7x7 / stride-4 patch-embed
    convolution (mix_transformer.py:169-183) runs with a padded weight copy on the LDS-DMA GEMM path; its weight gradient goes
    through a padded shadow whose drain keeps the 3 real channels.  Output, LayerNorm and every parameter gradient == torch."""
    import torch.nn.functional as Fn
    import cmda_amd.runtime as rt_
    from cmda_amd import ops as ops_
    from cmda_amd.backbones import OverlapPatchEmbed
    torch.manual_seed(3)
    rt_.set_compute_dtype(torch.bfloat16)
    try:
        B, H, W = 2, 32, 40
        pe = OverlapPatchEmbed(patch_size=7, stride=4, in_chans=3, embed_dim=64)
        with torch.no_grad():
            pe.proj.weight.copy_(torch.randn_like(pe.proj.weight) * 0.1)
            pe.proj.bias.copy_(torch.randn(64) * 0.1)
            pe.norm.weight.copy_(1 + 0.1 * torch.randn(64))
            pe.norm.bias.copy_(0.1 * torch.randn(64))
        pe.to(tgt.device)
        img = torch.randn(B, 3, H, W)
        cp = rt_.conv_channel_pad(3)
        assert cp == 8
        x = torch.empty(B * H * W, cp, dtype=torch.bfloat16, device=tgt.device)
        ops_.nchw_to_nhwc_pad(tgt.to(img), x, B, 3, H * W, cp)
        assert torch.equal(x[:, :3].float().cpu().view(B, H, W, 3), img.bfloat16().float().permute(0, 2, 3, 1)) and float(x[:, 3:].abs().sum()) == 0
        yn, OH, OW, sv = pe.fwd(x, B, H, W)
        g = torch.randn(B * OH * OW, 64)
        for p_ in pe.parameters():
            p_.grad = None
        pe.bwd(sv, tgt.to(g.bfloat16()), B, need_dx=False)
        # torch reference on the bf16-rounded operands
        xr = img.bfloat16().float()
        w = pe.proj.weight.detach().cpu().bfloat16().float().requires_grad_(True)
        b = pe.proj.bias.detach().cpu().clone().requires_grad_(True)
        gw = pe.norm.weight.detach().cpu().clone().requires_grad_(True)
        gb = pe.norm.bias.detach().cpu().clone().requires_grad_(True)
        y = Fn.conv2d(xr, w, b, 4, 3)
        assert y.shape[2:] == (OH, OW)
        yl = y.permute(0, 2, 3, 1).reshape(-1, 64)
        ref = Fn.layer_norm(yl.bfloat16().float(), (64,), gw, gb, 1e-5)
        ref.backward(g.bfloat16().float())
        assert_close(yn, ref, 3e-2, name='patch embed + LN (padded channels)')
        assert_close(pe.proj.weight.grad, w.grad, 3e-2, name='dW through the padded shadow')
        assert_close(pe.proj.bias.grad, b.grad, 3e-2, name='db')
        assert_close(pe.norm.weight.grad, gw.grad, 3e-2, name='dgamma')
    finally:
        rt_.set_compute_dtype(torch.float32)
