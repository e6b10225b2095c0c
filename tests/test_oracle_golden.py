"""Pins the oracle (oracle/*.py, CPU restatement) to the reference: every case compares against outputs produced by
the reference's own unmodified modules (tests/golden/*.npz, see tests/golden/make_golden.py)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
from weights import sample_grad, seeded_fill, seeded_randn  # noqa: E402

from oracle import cyclegan as ocg  # noqa: E402
from oracle import fusion as ofu  # noqa: E402
from oracle import head as ohd  # noqa: E402
from oracle import mit as omit  # noqa: E402
from oracle import segmentor as oseg  # noqa: E402
from oracle import uda as ouda  # noqa: E402
from conftest import assert_close, assert_close_fingerprint  # noqa: E402

torch.set_num_threads(8)


def gold(name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(HERE, 'golden', name + '.npz')).items()}


def check_grads(module, g, rtol, n=2048):
    seen = 0
    for name, p in module.named_parameters():
        key = 'grad.' + name
        if key in g:
            assert p.grad is not None, name
            assert_close_fingerprint(sample_grad(p.grad, n), g[key], rtol, atol=1e-6, name=key)
            seen += 1
    assert seen == sum(k.startswith('grad.') for k in g)


BLOCK_CFGS = {'s1': (64, 1, 8, 16, 16), 's2': (128, 2, 4, 8, 16), 's3': (320, 5, 2, 8, 8), 's4': (512, 8, 1, 4, 4),
              'f1': (128, 1, 4, 8, 8)}


@pytest.mark.parametrize('tag', list(BLOCK_CFGS))
def test_block(tag):
    dim, heads, sr, H, W = BLOCK_CFGS[tag]
    g = gold('block_' + tag)
    m = seeded_fill(omit.Block(dim, heads, 4, True, 0.0, sr), 11).train()
    x = seeded_randn((2, H * W, dim), 11, 'x').requires_grad_(True)
    y = m(x, H, W)
    y.backward(seeded_randn(y.shape, 11, 'dy'))
    assert_close(y, g['y'], 1e-5, name='y')
    assert_close(x.grad, g['dx'], 1e-5, name='dx')
    check_grads(m, g, 2e-5)


def test_mit_b5_eval():
    g = gold('mit_b5_64')
    m = seeded_fill(omit.mit_b5(), 21).eval()
    with torch.no_grad():
        outs = m(seeded_randn((1, 3, 64, 64), 21, 'img'))
    for i, o in enumerate(outs):
        assert_close(o, g[f'out{i}'], 2e-5, name=f'out{i}')


def test_mit_small_train():
    g = gold('mit_small_train')
    m = seeded_fill(omit.MixVisionTransformer(depths=(1, 1, 1, 1), drop_path_rate=0.0), 22).train()
    outs = m(seeded_randn((2, 3, 64, 96), 22, 'img'))
    sum((o * seeded_randn(o.shape, 22, f'dy{i}')).sum() for i, o in enumerate(outs)).backward()
    for i, o in enumerate(outs):
        assert_close(o, g[f'out{i}'], 1e-5, name=f'out{i}')
    check_grads(m, g, 5e-5)


def feats(B, H, W, seed, tag):
    return [seeded_randn((B, c, H // s, W // s), seed, f'{tag}{i}').requires_grad_(True)
            for i, (c, s) in enumerate(zip([64, 128, 320, 512], [4, 8, 16, 32]))]


def test_head_train():
    g = gold('head_train')
    head = seeded_fill(ohd.DAFormerHead(dropout_ratio=0.0), 31).train()
    fs = feats(2, 64, 96, 31, 'f')
    losses, logits = head.forward_train(fs, g['gt'], g['weight'])
    (losses['loss_seg'] * 1.7).backward()
    assert_close(logits, g['logits'], 1e-5, name='logits')
    assert_close(losses['loss_seg'], g['loss_seg'], 1e-6, name='loss')
    assert_close(losses['acc_seg'], g['acc_seg'], 1e-6, name='acc')
    for i, f in enumerate(fs):
        assert_close(f.grad, g[f'dfeat{i}'], 2e-5, atol=1e-9, name=f'dfeat{i}')
    check_grads(head, g, 5e-5)
    for k, v in head.state_dict().items():
        if 'running' in k:
            assert_close(v, g['bn.' + k], 1e-5, name=k)


def test_head_fusion_train():
    g = gold('head_fusion_train')
    head = seeded_fill(ohd.DAFormerHeadFusion(dropout_ratio=0.0, share_decoder=True), 41).train()
    with open(os.path.join(HERE, 'golden', 'head_fusion_keys.json')) as f:
        assert sorted(head.state_dict().keys()) == json.load(f)
    inputs = {k: feats(1, 64, 64, 41, k) for k in ('f_image', 'f_events', 'f_fusion', 'f_img_self_res')}
    cfg = dict(loss_weight={'image': 0.5, 'events': 0.5, 'fusion': 0.5, 'img_self_res': 0.25})
    losses, logits = head.forward_train(inputs, g['gt'], None, cfg)
    losses['loss_seg'].backward()
    assert_close(losses['loss_seg'], g['loss_seg'], 1e-6, name='loss')
    assert_close(losses['acc_seg'], g['acc_seg'], 1e-6, name='acc')
    for k, v in logits.items():
        assert_close(v, g[k], 1e-5, name=k)
    for k, fs in inputs.items():
        for i, f in enumerate(fs):
            assert_close(f.grad, g[f'd{k}{i}'], 2e-5, atol=1e-9, name=f'd{k}{i}')
    check_grads(head, g, 5e-5)


def test_isr():
    g = gold('isr')
    img = g['img'].numpy()
    gray = ouda.pil_luma(img)
    assert np.array_equal(gray, g['gray'].numpy())
    params = {'dsec': dict(val_range=[0.01, 1.01], threshold=0.005, clip_range=0.1, shift_pixel=1),
              'dz': dict(val_range=[1, 100], threshold=0.01, clip_range=0.1, shift_pixel=3)}
    for pn, p in params.items():
        for d in ('rightdown', 'rightup', 'leftdown', 'leftup', 'all'):
            out = ouda.image_change(gray, shift_direction=d, **p)
            assert torch.equal(out, g[f'{pn}_{d}']), (pn, d)


def test_voxel():
    g = gold('voxel')
    for bins in (1, 5):
        vg = ouda.events_to_voxel_grid(g[f't{bins}'], g[f'x{bins}'], g[f'y{bins}'], g[f'p{bins}'], 64, 48, bins)
        assert_close(vg, g[f'vg{bins}'], 1e-6, name='voxel grid')
        nrm = ouda.events_norm(g[f'vg{bins}'].clone(), clip_range=(5000 / 500000) * 1.5 * 100)
        assert_close(nrm, g[f'norm{bins}'], 1e-6, name='events_norm')


def test_classmix():
    g = gold('classmix')
    lab, cls = g['label'], g['classes']
    rng = np.random.RandomState(71)
    chosen = ouda.choose_classes(lab, rng)
    for i in range(lab.shape[0]):
        assert torch.equal(chosen[i], cls[i][cls[i] >= 0])
        m = ouda.class_mask(lab[i], chosen[i])
        assert torch.equal(m, g['masks'][i])
        assert torch.equal(ouda.one_mix(m, g['a'][i], g['b'][i]), g['mixed'][i])
        assert torch.equal(ouda.one_mix(m, lab[i][0], g['pl'][i]), g['mixed_label'][i])


def test_generator():
    g = gold('generator')
    G = seeded_fill(ocg.ResnetGenerator(), 81).eval()
    with open(os.path.join(HERE, 'golden', 'generator_keys.json')) as f:
        assert sorted(G.state_dict().keys()) == json.load(f)
    with torch.no_grad():
        y = G(seeded_randn((2, 1, 32, 48), 81, 'x'))
    assert_close(y, g['y'], 1e-5, name='generator')


@pytest.mark.parametrize('name,cls', [('avg', ofu.AttentionAvgFusion), ('cat', ofu.AttentionFusion)])
def test_fusion_modules(name, cls):
    g = gold('fusion_' + name)
    m = seeded_fill(cls(drop_path_rate=0.0), 91).train()
    fi = [f.detach() for f in feats(1, 64, 64, 91, 'i')]
    fe = [f.detach() for f in feats(1, 64, 64, 91, 'e')]
    for i, o in enumerate(m(fi, fe)):
        assert_close(o, g[f'out{i}'], 2e-5, name=f'out{i}')


def test_segmentor_train_and_teacher():
    g = gold('segmentor_train')
    model = oseg.FusionEncoderDecoder(
        backbone_image=omit.mit_b5(drop_path_rate=0.0), backbone_events=omit.mit_b5(drop_path_rate=0.0),
        fusion_module=ofu.AttentionAvgFusion(drop_path_rate=0.0),
        decode_head=ohd.DAFormerHeadFusion(dropout_ratio=0.0, share_decoder=True))
    with open(os.path.join(HERE, 'golden', 'segmentor_keys.json')) as f:
        assert sorted(model.state_dict().keys()) == json.load(f)
    seeded_fill(model, 101).train()
    inputs = {k: seeded_randn((1, 3, 64, 64), 101, k) for k in ('image', 'events', 'img_self_res')}
    fcfg = dict(loss_weight={'image': 0.5, 'events': 0.5, 'fusion': 0.5, 'img_self_res': 0.25})
    losses, pred = model.forward_train(inputs, g['gt'], return_feat=True, cfg=fcfg)
    losses['decode.loss_seg'].backward()
    assert_close(losses['decode.loss_seg'], g['loss_seg'], 2e-6, name='loss')
    assert_close(losses['decode.acc_seg'], g['acc_seg'], 1e-6, name='acc')
    for k, v in pred.items():
        assert_close(v, g[k], 5e-5, name=k)
    # fp32 round-off through 2 x 52 blocks: attention-q gradients (softmax-Jacobian cancellation, Nkv = 4 here) differ by up
    # to ~1% between two fp32 CPU evaluations that only differ in op order; everything else agrees to ~1e-4.
    check_grads(model, g, 2e-2, n=96)
    gt = gold('segmentor_teacher')
    model.eval()
    with torch.no_grad():
        out = model.encode_decode(inputs['image'], inputs['events'], output_features=True, test_cfg=fcfg)
    for k, v in gt.items():
        assert_close(out[k], v, 5e-5, name=k)


def test_pseudo_label_exact_upsample_matches_interpolate():
    torch.manual_seed(0)
    lg = torch.randn(2, 19, 16, 24) * 3
    a = ouda.upsample_exact(lg, (64, 96))
    b = torch.nn.functional.interpolate(lg, size=(64, 96), mode='bilinear', align_corners=False)
    assert_close(a, b, 1e-6, name='upsample_exact')
    lab, prob, w, cnt = ouda.pseudo_labels(lg, (64, 96), 0.968)
    pr, lr = torch.softmax(b, 1).max(1)
    assert (lab != lr).float().mean().item() < 1e-4
    assert_close(prob, pr, 1e-5, name='prob')


def test_schedule_and_param_groups():
    assert abs(ouda.poly_warm_lr(6e-5, 0) - 6e-5 * 1e-6) < 1e-15
    assert abs(ouda.poly_warm_lr(6e-5, 1500) - 6e-5 * (1 - 1500 / 40000)) < 1e-12
    keys = dict(head=dict(lr_mult=10.0), pos_block=dict(decay_mult=0.0), norm=dict(decay_mult=0.0))
    assert ouda.param_group_options('model.decode_head.conv_seg.weight', 6e-5, 0.01, keys) == pytest.approx((6e-4, 0.01))
    assert ouda.param_group_options('model.backbone_image.block1.0.norm1.weight', 6e-5, 0.01, keys) == (6e-5, 0.0)
    assert ouda.param_group_options('model.backbone_image.block1.0.attn.q.weight', 6e-5, 0.01, keys) == (6e-5, 0.01)


# ---------------------------------------------------------------------------------------------------------------------------
# The step COMPOSITION against the reference's own DACS.train_step (tests/golden/dacs_step.npz written by
# make_golden.py::dacs_step from mmseg/models/uda/dacs.py:274-315,357-860 and _init_ema_weights / _update_ema :250-272)
from weights import DACS_CH, DACS_DIMS, DACS_SEEDS, DACS_SEG_SCALE, dacs_batch  # noqa: E402

DACS_FCFG = dict(loss_weight={'image': 0.5, 'events': 0.5, 'fusion': 0.5, 'img_self_res': 0.25}, gradual_rate=0.0)
DACS_ISR = dict(val_range=[0.01, 1.01], _threshold=0.005, _clip_range=0.1, shift_pixel=1)


def dacs_fixture_draws(g, it):
    """the host decisions the reference made in iteration `it` (recorded in the fixture) as oracle.dacs_iter `draws`"""
    cj, bl, sigma = [float(v) for v in g[f'it{it}.gates']]
    return dict(choice=float(g[f'it{it}.choice']), color_jitter=cj, blur=bl, sigma=sigma, classes=[g[f'it{it}.classes']], jitter=None)


def dacs_fixture_models():
    from oracle import dacs_iter  # noqa: F401
    def net():
        return oseg.FusionEncoderDecoder(backbone_image=omit.MixVisionTransformer(embed_dims=DACS_DIMS, depths=[1, 1, 1, 1], drop_path_rate=0.0),
                                         backbone_events=omit.MixVisionTransformer(embed_dims=DACS_DIMS, depths=[1, 1, 1, 1], drop_path_rate=0.0),
                                         fusion_module=ofu.AttentionAvgFusion(in_channels=DACS_DIMS, drop_path_rate=0.0),
                                         decode_head=ohd.DAFormerHeadFusion(in_channels=DACS_DIMS, channels=DACS_CH, embed_dims=DACS_CH,
                                                                            dropout_ratio=0.0, share_decoder=True))
    student, teacher, G = net(), net(), ocg.ResnetGenerator().eval()
    seeded_fill(student, DACS_SEEDS['student']).train()
    seeded_fill(teacher, DACS_SEEDS['teacher']).train()
    seeded_fill(G, DACS_SEEDS['generator'])
    with torch.no_grad():
        student.decode_head.conv_seg.weight.mul_(DACS_SEG_SCALE)
    return student, teacher, G


def check_dacs_fixture_iteration(g, it, o, named_params, named_ema, named_buffers, tol=2e-4, grad_tol=2e-3, n=24):
    """one iteration's observables against the fixture: o = dict with the oracle.dacs_iter keys"""
    ref_l = g[f'it{it}.losses']
    got_l = torch.tensor([float(o['decode.loss_seg']), float(o['decode.acc_seg']), float(o['mix.decode.loss_seg']), float(o['mix.decode.acc_seg'])])
    assert_close(got_l[[0, 2]], ref_l[[0, 2]].float(), tol, name=f'it{it} losses')
    assert_close(got_l[[1, 3]], ref_l[[1, 3]].float(), 1e-3, name=f'it{it} accuracies')
    agree = (o['pseudo_label'].cpu().to(torch.uint8) == g[f'it{it}.pseudo_label']).float().mean().item()
    assert agree >= 0.9995, f'it{it} pseudo-label agreement {agree}'
    assert abs(int(o['pseudo_count']) - int(g[f'it{it}.pseudo_conf'])) <= 40, (int(o['pseudo_count']), int(g[f'it{it}.pseudo_conf']))
    assert_close(o['mixed_img'].cpu()[..., ::4, ::4], g[f'it{it}.mixed_img_s'], 1e-6, name=f'it{it} mixed image')
    assert_close(o['mixed_events'].cpu()[:, :1, ::4, ::4], g[f'it{it}.mixed_events_s'], tol, atol=1e-5, name=f'it{it} mixed events')
    assert_close(o['mixed_isr'].cpu()[:, :1, ::2, ::2], g[f'it{it}.mixed_isr_s'].float(), 1e-3, name=f'it{it} mixed ISR')
    same = (o['mixed_lbl'].cpu().to(torch.uint8) == g[f'it{it}.mixed_lbl']).float().mean().item()
    assert same >= 0.9995, f'it{it} mixed-label agreement {same}'
    assert_close(o['mixed_weight'].cpu()[..., ::8, ::8], g[f'it{it}.mixed_weight_s'], 1e-3, name=f'it{it} mixed pseudo-weight')
    seen = 0
    for k, p in named_params:
        assert_close_fingerprint(sample_grad(p.grad, n), g[f'it{it}.grad.{k}'], grad_tol, atol=1e-6, name=f'it{it} grad {k}')
        seen += 1
    assert seen == sum(k.startswith(f'it{it}.grad.') for k in g)
    for k, p in named_ema:
        assert_close(sample_grad(p.data, n), g[f'it{it}.ema.{k}'], 1e-5, atol=3e-4 if it else 1e-7, name=f'it{it} ema {k}')
    for k, b in named_buffers:
        if f'it{it}.bn.{k}' in g:
            assert_close(b, g[f'it{it}.bn.{k}'], 2e-4 if it == 0 else 2e-3, atol=1e-5, name=f'it{it} {k}')


@pytest.mark.slow
def test_dacs_step_against_reference_train_step():
    """oracle/dacs_iter.py + torch AdamW, three iterations (local_iter 0, 1, 2) with the reference's recorded draws, against what the
    reference's own DACS.train_step produced: losses, pseudo-labels, confident-pixel count, mixed image / events / ISR / label /
    pseudo-weight, accumulated gradients, EMA teacher, student after the optimizer step, BatchNorm running statistics; then
    _update_ema(1500)"""
    from oracle import dacs_iter
    g = gold('dacs_step')
    student, teacher, G = dacs_fixture_models()
    opt = torch.optim.AdamW(student.parameters(), lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01)
    src, tg = dacs_batch()
    for it in range(3):
        opt.zero_grad()
        o = dacs_iter.dacs_iteration(student, teacher, G, {k: v.clone() for k, v in src.items()}, {k: v.clone() for k, v in tg.items()},
                                     local_iter=it, forward_cfg=DACS_FCFG, isr_parms=DACS_ISR, shift_type='random',
                                     draws=dacs_fixture_draws(g, it))
        assert o['use_events'] == (float(g[f'it{it}.choice']) > 0.5)
        # iteration 0 is compared at fp32 round-off; behind the first optimizer step the two runs differ by AdamW's +-lr noise on the
        # zero-gradient parameters, which moves a handful of pseudo-labels and with them the pixel-summed gradients
        check_dacs_fixture_iteration(g, it, o, list(student.named_parameters()), list(teacher.named_parameters()),
                                     list(student.named_buffers()), grad_tol=2e-3 if it == 0 else 6e-2)
        opt.step()
        for k, p in student.named_parameters():
            # (AdamW turns a gradient that is zero up to round-off -- the key half of kv.bias cannot move the softmax -- into +-lr steps of
            # arbitrary sign: the absolute term is a few lr = 6e-5 on the fingerprint's sums)
            assert_close(sample_grad(p.data, 24), g[f'it{it}.param.{k}'], 1e-5, atol=3e-4, name=f'it{it} param {k}')
    ouda.update_ema(list(teacher.parameters()), list(student.parameters()), 1500, 0.999)
    for k, p in teacher.named_parameters():
        assert_close(sample_grad(p.data, 24), g[f'ema1500.{k}'], 1e-5, atol=3e-4, name=f'ema1500 {k}')
