"""Generates tests/golden/*.npz from the reference's OWN modules (imported unmodified from /root/reference through
tests/golden/ref_shim.py).  Runs only in the authoring container; the fixtures (inputs/seeds + expected outputs) are
committed, the reference never ships.  Usage: python tests/golden/make_golden.py [case ...]
"""
import json
import os
import sys
from functools import partial

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
from weights import (DACS_CH, DACS_DIMS, DACS_SEG_SCALE, dacs_batch, sample_grad, seeded_fill,  # noqa: E402
                     seeded_randn)

torch.set_num_threads(8)
ns = ref_shim.load_hotpath()
CASES = {}


def case(fn):
    CASES[fn.__name__] = fn
    return fn


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('wrote', name, {k: tuple(v.shape) for k, v in out.items()})


def param_grads(module, n=2048):
    return {'grad.' + k: sample_grad(p.grad, n) for k, p in module.named_parameters() if p.grad is not None}


BLOCK_CFGS = {'s1': (64, 1, 8, 16, 16), 's2': (128, 2, 4, 8, 16), 's3': (320, 5, 2, 8, 8), 's4': (512, 8, 1, 4, 4),
              'f1': (128, 1, 4, 8, 8)}


@case
def block():
    for tag, (dim, heads, sr, H, W) in BLOCK_CFGS.items():
        m = ns.mit.Block(dim=dim, num_heads=heads, mlp_ratio=4, qkv_bias=True, drop_path=0.0,
                         norm_layer=partial(nn.LayerNorm, eps=1e-6), sr_ratio=sr)
        seeded_fill(m, 11)
        m.train()
        x = seeded_randn((2, H * W, dim), 11, 'x').requires_grad_(True)
        y = m(x, H, W)
        y.backward(seeded_randn(y.shape, 11, 'dy'))
        save(f'block_{tag}', y=y, dx=x.grad, **param_grads(m))


@case
def mit_b5_64():
    m = ns.mit.mit_b5(style='pytorch', in_chans=3)
    seeded_fill(m, 21)
    m.eval()
    with torch.no_grad():
        outs = m(seeded_randn((1, 3, 64, 64), 21, 'img'))
    save('mit_b5_64', **{f'out{i}': o for i, o in enumerate(outs)})


@case
def mit_small_train():
    m = ns.mit.MixVisionTransformer(patch_size=4, embed_dims=[64, 128, 320, 512], num_heads=[1, 2, 5, 8],
                                    mlp_ratios=[4, 4, 4, 4], qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                    depths=[1, 1, 1, 1], sr_ratios=[8, 4, 2, 1], drop_path_rate=0.0)
    seeded_fill(m, 22)
    m.train()
    outs = m(seeded_randn((2, 3, 64, 96), 22, 'img'))
    loss = sum((o * seeded_randn(o.shape, 22, f'dy{i}')).sum() for i, o in enumerate(outs))
    loss.backward()
    save('mit_small_train', **{f'out{i}': o for i, o in enumerate(outs)}, **param_grads(m))


HEAD_CFG = dict(in_channels=[64, 128, 320, 512], in_index=[0, 1, 2, 3], channels=256, dropout_ratio=0.0, num_classes=19,
                norm_cfg=dict(type='BN', requires_grad=True), align_corners=False,
                loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0))


def _decoder_params(**extra):
    d = dict(embed_dims=256, embed_cfg=dict(type='mlp', act_cfg=None, norm_cfg=None),
             embed_neck_cfg=dict(type='mlp', act_cfg=None, norm_cfg=None),
             fusion_cfg=dict(type='aspp', sep=True, dilations=(1, 6, 12, 18), pool=False, act_cfg=dict(type='ReLU'),
                             norm_cfg=dict(type='BN', requires_grad=True)))
    d.update(extra)
    return d


def _feats(B, H, W, seed, tag):
    return [seeded_randn((B, c, H // s, W // s), seed, f'{tag}{i}').requires_grad_(True)
            for i, (c, s) in enumerate(zip([64, 128, 320, 512], [4, 8, 16, 32]))]


def _label(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    lab = torch.randint(0, 19, (B, 1, H // 8, W // 8), generator=g).repeat_interleave(8, 2).repeat_interleave(8, 3)
    lab[torch.rand((B, 1, H, W), generator=g) < 0.05] = 255
    return lab


@case
def head_train():
    head = ns.daformer_head.DAFormerHead(**HEAD_CFG, decoder_params=_decoder_params())
    seeded_fill(head, 31)
    head.train()
    B, H, W = 2, 64, 96
    feats = _feats(B, H, W, 31, 'f')
    gt = _label(B, H, W, 31)
    weight = torch.rand((B, H, W), generator=torch.Generator().manual_seed(32))
    losses, logits = head.forward_train(feats, None, gt, None, weight)
    (losses['loss_seg'] * 1.7).backward()
    bn = {k: v for k, v in head.state_dict().items() if 'running' in k}
    save('head_train', logits=logits, loss_seg=losses['loss_seg'], acc_seg=losses['acc_seg'], gt=gt, weight=weight,
         **{f'dfeat{i}': f.grad for i, f in enumerate(feats)}, **param_grads(head), **{'bn.' + k: v for k, v in bn.items()})


@case
def head_fusion_train():
    head = ns.daformer_head.DAFormerHeadFusion(
        **HEAD_CFG, decoder_params=_decoder_params(train_type='cs2dsec_image+events_together', share_decoder=True))
    seeded_fill(head, 41)
    head.train()
    B, H, W = 1, 64, 64
    inputs = {k: _feats(B, H, W, 41, k) for k in ('f_image', 'f_events', 'f_fusion', 'f_img_self_res')}
    gt = _label(B, H, W, 41)
    cfg = dict(loss_weight={'image': 0.5, 'events': 0.5, 'fusion': 0.5, 'img_self_res': 0.25}, gradual_rate=0.0)
    losses, logits = head.forward_train(inputs, None, gt, None, None, cfg)
    losses['loss_seg'].backward()
    # the shared decoder's BatchNorm layers have now seen image, events, fusion, ISR features in that order (four updates)
    bn = {k: v for k, v in head.state_dict().items() if 'running' in k and k.startswith('fuse_layer_image')}
    save('head_fusion_train', loss_seg=losses['loss_seg'], acc_seg=losses['acc_seg'], gt=gt,
         **{k: v for k, v in logits.items()}, **{f'd{k}{i}': f.grad for k, fs in inputs.items() for i, f in enumerate(fs)},
         **param_grads(head), **{'bn.' + k: v for k, v in bn.items()})
    with open(os.path.join(HERE, 'head_fusion_keys.json'), 'w') as f:
        json.dump(sorted(head.state_dict().keys()), f, indent=0)


@case
def isr():
    from PIL import Image
    g = torch.Generator().manual_seed(51)
    H, W = 40, 56
    base = torch.rand((H // 4, W // 4, 3), generator=g).repeat_interleave(4, 0).repeat_interleave(4, 1)
    img = ((base * 0.8 + 0.2 * torch.rand((H, W, 3), generator=g)) * 255).to(torch.uint8).numpy()
    pil = Image.fromarray(img)
    out = {'img': img, 'gray': np.array(pil.convert('L'))}
    params = {'dsec': dict(val_range=[0.01, 1.01], _threshold=0.005, _clip_range=0.1, shift_pixel=1),
              'dz': dict(val_range=[1, 100], _threshold=0.01, _clip_range=0.1, shift_pixel=3)}
    for pn, p in params.items():
        for d in ('rightdown', 'rightup', 'leftdown', 'leftup', 'all'):
            out[f'{pn}_{d}'] = ns.ds_utils.get_image_change_from_pil(pil, width=W, height=H, auto_threshold=None,
                                                                       shift_direction=d, **p)
    save('isr', **out)


@case
def voxel():
    dsec = ref_shim.load('mmseg.datasets.dsec')
    g = torch.Generator().manual_seed(61)
    N, W, H = 5000, 64, 48
    out = {}
    for bins in (1, 5):
        t = torch.sort(torch.rand(N, generator=g))[0] * 1e5
        x = torch.rand(N, generator=g) * (W + 2) - 1
        y = torch.rand(N, generator=g) * (H + 2) - 1
        x, y = x.clamp(0, W - 0.001), y.clamp(0, H - 0.001)
        p = (torch.rand(N, generator=g) > 0.5).float()
        vg = dsec.events_to_voxel_grid(t, x, y, p, W, H, bins, normalize_flag=False)
        out.update({f't{bins}': t, f'x{bins}': x, f'y{bins}': y, f'p{bins}': p, f'vg{bins}': vg,
                    f'norm{bins}': dsec.events_norm(vg.clone(), clip_range=(N / 500000) * 1.5 * 100, final_range=1.0,
                                                     enforce_no_events_zero=True)})
    save('voxel', **out)


@case
def pipeline():
    """Loader-side preprocessing on synthetic raw data, step by step as the reference's loaders do it:
      target  mmseg/datasets/dsec.py:189-339 (__getitem__: crop -> flip -> PIL resize -> transform; real-time ISR; events crop /
              flip / F.interpolate / x3) and :341-366 (get_events_vg) -- the file reads are replaced by the synthetic arrays, the
              arithmetic is the reference's own (events_to_voxel_grid, events_norm, get_image_change_from_pil imported unmodified;
              PIL and torch ops called exactly as the loader calls them; torchvision ToTensor / Normalize restated: absent here);
      source  mmseg/datasets/cityscapes_ic.py:147-210 (resize -> crop -> flip of image and time residual, ISR of the cropped image);
      time residual  create_cityscapes_image_change.py:16-35 get_image_change with its __main__ constants (:169-172)."""
    import importlib.util
    import torch.nn.functional as F
    from PIL import Image, ImageOps
    dsec = ref_shim.load('mmseg.datasets.dsec')
    g = torch.Generator().manual_seed(81)
    mean, std = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)

    def to_tensor_norm(pil):           # torchvision ToTensor + Normalize (third-party, restated)
        t = torch.from_numpy(np.asarray(pil).copy()).permute(2, 0, 1).float().div(255)
        return (t - mean) / std

    def smooth_u8(h, w, c):            # blocky + noisy synthetic frame
        base = torch.rand((h // 4 + 1, w // 4 + 1, c), generator=g).repeat_interleave(4, 0).repeat_interleave(4, 1)[:h, :w]
        return ((base * 0.8 + 0.2 * torch.rand((h, w, c), generator=g)) * 255).to(torch.uint8).numpy()
    out = {}
    # ---- target (DSEC), scaled-down geometry: frame 60x80, crop 40x40, resize to 52x52
    FH, FW, CROP, RES = 60, 80, 40, 52
    isr_p = dict(val_range=[0.01, 1.01], _threshold=0.005, _clip_range=0.1, shift_pixel=1)
    frame = smooth_u8(FH, FW, 3)
    N = 4000
    t_us = torch.sort(torch.randint(0, 50000, (N,), generator=g))[0].numpy().astype(np.int64) + 1234567
    ex = torch.randint(0, FW, (N,), generator=g).numpy().astype(np.int32)
    ey = torch.randint(0, FH, (N,), generator=g).numpy().astype(np.int32)
    ep = torch.randint(0, 2, (N,), generator=g).numpy().astype(np.uint8)
    yy, xx = np.meshgrid(np.arange(FH, dtype=np.float32), np.arange(FW, dtype=np.float32), indexing='ij')
    rect = np.stack([np.clip(xx + 0.6 * np.sin(yy / 7.0), 0, FW - 1.001), np.clip(yy + 0.4 * np.cos(xx / 5.0), 0, FH - 1.001)],
                    axis=-1).astype(np.float32)                        # rectify_map[y, x] = (x_rect, y_rect)
    out.update(t_frame=frame, t_t=t_us, t_x=ex, t_y=ey, t_p=ep, t_rect=rect)
    # get_events_vg :341-366
    et = (t_us - t_us[0]).astype('float32')
    et = torch.from_numpy(et / et[-1])
    pol = torch.from_numpy(ep.astype('float32'))
    xy = rect[ey, ex]
    erx, ery = torch.from_numpy(xy[:, 0].astype('float32')), torch.from_numpy(xy[:, 1].astype('float32'))
    vg = dsec.events_to_voxel_grid(et, erx, ery, pol, FW, FH, num_bins=1, normalize_flag=False)
    clip = (N - 1) / 500000 * 1.5 * 100     # events_finish - events_start = N-1; x100 keeps the tiny synthetic grid off the clip
    vg = dsec.events_norm(vg, clip_range=clip, final_range=1.0, enforce_no_events_zero=True)
    out['t_vg'] = vg
    for tag, (x, y, flip, direction) in {'a': (7, 3, False, 'rightdown'), 'b': (40, 20, True, 'leftup')}.items():
        pil = Image.fromarray(frame).convert('RGB')
        pil = pil.crop(box=(x, y, x + CROP, y + CROP))
        if flip:
            pil = ImageOps.mirror(pil)       # RandomHorizontalFlip(p=1)
        pil = pil.resize(size=(RES, RES), resample=Image.BILINEAR)
        out[f't_{tag}_warp_u8'] = np.asarray(pil).copy()
        out[f't_{tag}_warp_image'] = to_tensor_norm(pil)
        isr = ns.ds_utils.get_image_change_from_pil(pil, width=RES, height=RES, shift_direction=direction, **isr_p)
        out[f't_{tag}_isr'] = isr.repeat(3, 1, 1)
        ev = vg[:, y: y + CROP, x: x + CROP]
        if flip:
            ev = torch.flip(ev, dims=[-1])
        ev = F.interpolate(ev[None], size=(RES, RES), mode='bilinear', align_corners=False)[0]
        out[f't_{tag}_events_vg'] = ev.repeat(3, 1, 1)
        out[f't_{tag}_params'] = np.array([x, y, int(flip)])
    # ---- source (Cityscapes), scaled down: frame 64x128 -> resize 64x32 (2:1, the antialiased branch) -> crop 32x32
    SH, SW = 64, 128
    f_now, f_prev = smooth_u8(SH, SW, 3), None
    f_prev = np.clip(f_now.astype(np.int32) + (torch.randn((SH, SW, 3), generator=g) * 18).numpy().astype(np.int32), 0, 255).astype(np.uint8)
    out.update(s_now=f_now, s_prev=f_prev)
    sys.modules['mmseg.models.cyclegan'].define_G = ns.cyclegan.define_G   # the script's only package-level import
    spec = importlib.util.spec_from_file_location('ccic', '/root/reference/create_cityscapes_image_change.py')
    ccic = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ccic)
    ccic.log_add, ccic.threshold, ccic.clip_range = 50, 0.1, 0.8       # its __main__ constants (:169-172)
    tr_pil = ccic.get_image_change(Image.fromarray(f_now).convert('L'), Image.fromarray(f_prev).convert('L'))
    out['s_time_res_u8'] = np.asarray(tr_pil).copy()
    for tag, (x, y, flip) in {'a': (5, 0, False), 'b': (31, 0, True)}.items():
        raw = Image.fromarray(f_now).convert('RGB')
        rs = raw.resize(size=(SW // 2, SH // 2), resample=Image.BILINEAR)
        cr = rs.crop(box=(x, y, x + 32, y + 32))
        if flip:
            cr = ImageOps.mirror(cr)
        out[f's_{tag}_image'] = to_tensor_norm(cr)
        isr = ns.ds_utils.get_image_change_from_pil(cr, width=32, height=32, shift_direction='rightdown', **isr_p)
        out[f's_{tag}_isr'] = isr.repeat(3, 1, 1)
        itr = tr_pil.convert('L').resize(size=(SW // 2, SH // 2), resample=Image.BILINEAR)
        itr = itr.crop(box=(x, y, x + 32, y + 32))
        if flip:
            itr = ImageOps.mirror(itr)
        itr = np.asarray(itr, dtype=np.float32)
        out[f's_{tag}_img_time_res'] = ((torch.from_numpy(itr)[None] / 255.0 - 0.5) / 0.5).repeat(3, 1, 1)
        out[f's_{tag}_params'] = np.array([x, y, int(flip)])
    save('pipeline', **out)


@case
def metrics():
    """mmseg/core/evaluation/metrics.py: intersect_and_union :27-87, total_intersect_and_union :90-126, eval_metrics :259-328
    (mIoU / mDice / mFscore, nan_to_num) on seeded label maps (half the DSEC evaluation size), incl. ignore pixels, a class absent from
    prediction and ground truth (NaN), reduce_zero_label and label_map."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_metrics', '/root/reference/mmseg/core/evaluation/metrics.py')
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rng = np.random.RandomState(91)
    nc = 19
    preds = [rng.randint(0, nc, (220, 320)) for _ in range(3)]
    gts = [rng.randint(0, nc, (220, 320)) for _ in range(3)]
    for g_, p_ in zip(gts, preds):
        g_[rng.rand(*g_.shape) < 0.1] = 255
        g_[g_ == 7] = 3
        p_[p_ == 7] = 2
    out = {f'pred{i}': p_.astype(np.uint8) for i, p_ in enumerate(preds)}
    out.update({f'gt{i}': g_.astype(np.uint8) for i, g_ in enumerate(gts)})
    for k, v in zip(('inter', 'union', 'area_pred', 'area_label'), m.total_intersect_and_union(preds, gts, nc, 255)):
        out['tot_' + k] = v
    for k, v in zip(('inter', 'union', 'area_pred', 'area_label'), m.intersect_and_union(preds[0], gts[0], nc, 255)):
        out['one_' + k] = v
    r = m.eval_metrics(preds, gts, nc, 255, metrics=['mIoU', 'mDice'])
    out.update({'m_' + k: v for k, v in r.items()})
    r0 = m.eval_metrics(preds, gts, nc, 255, metrics=['mIoU'], nan_to_num=0)
    out.update({'m0_' + k: v for k, v in r0.items()})
    rz = m.intersect_and_union(preds[1], gts[1], nc - 1, 255, reduce_zero_label=True)
    out.update({f'rz_{k}': v for k, v in zip(('inter', 'union', 'area_pred', 'area_label'), rz)})
    lm = m.intersect_and_union(preds[2], gts[2], nc, 255, label_map={5: 4, 9: 255})
    out.update({f'lm_{k}': v for k, v in zip(('inter', 'union', 'area_pred', 'area_label'), lm)})
    save('metrics', **out)


@case
def classmix():
    tr = ns.dacs_transforms
    g = torch.Generator().manual_seed(71)
    B, H, W = 2, 24, 32
    lab = torch.randint(0, 6, (B, 1, H // 4, W // 4), generator=g).repeat_interleave(4, 2).repeat_interleave(4, 3)
    lab[0, 0, :4] = 255
    np.random.seed(71)
    masks = tr.get_class_masks(lab)
    np.random.seed(71)  # replay the draw to record the chosen classes
    classes = torch.unique(lab)
    n = classes.shape[0]
    chosen = [classes[torch.Tensor(np.random.choice(n, int((n + n % 2) / 2), replace=False)).long()] for _ in range(B)]
    a, b = torch.randn((B, 3, H, W), generator=g), torch.randn((B, 3, H, W), generator=g)
    pl = torch.randint(0, 19, (B, H, W), generator=g)
    mixed, mixed_l = [], []
    for i in range(B):
        d, _ = tr.one_mix(masks[i], data=torch.stack((a[i], b[i])))
        _, t = tr.one_mix(masks[i], target=torch.stack((lab[i][0], pl[i])))
        mixed.append(d), mixed_l.append(t)
    K = max(len(c) for c in chosen)
    cls = torch.full((B, K), -1, dtype=torch.long)
    for i, c in enumerate(chosen):
        cls[i, :len(c)] = c
    save('classmix', label=lab, a=a, b=b, pl=pl, classes=cls, masks=torch.cat(masks), mixed=torch.cat(mixed),
         mixed_label=torch.cat(mixed_l))


@case
def generator():
    G = ns.cyclegan.ResnetGenerator(1, 1, 64, norm_layer=partial(nn.InstanceNorm2d, affine=False, track_running_stats=False),
                                    use_dropout=False, n_blocks=9)
    seeded_fill(G, 81)
    G.eval()
    with torch.no_grad():
        y = G(seeded_randn((2, 1, 32, 48), 81, 'x'))
    save('generator', y=y)
    with open(os.path.join(HERE, 'generator_keys.json'), 'w') as f:
        json.dump(sorted(G.state_dict().keys()), f, indent=0)


@case
def fusion_modules():
    feats_i = [f.detach() for f in _feats(1, 64, 64, 91, 'i')]
    feats_e = [f.detach() for f in _feats(1, 64, 64, 91, 'e')]
    for name, cls in (('avg', ns.avg_fusion.AttentionAvgFusion), ('cat', ns.att_fusion.AttentionFusion)):
        m = cls(drop_path_rate=0.0)
        seeded_fill(m, 91)
        m.train()
        outs = m(feats_i, feats_e)
        save(f'fusion_{name}', **{f'out{i}': o for i, o in enumerate(outs)})


@case
def segmentor_train():
    cfg = dict(type='FusionEncoderDecoder', pretrained=None,
               backbone_image=dict(type='mit_b5', style='pytorch', in_chans=3, drop_path_rate=0.0),
               backbone_events=dict(type='mit_b5', style='pytorch', in_chans=3, drop_path_rate=0.0),
               fusion_module=dict(type='AttentionAvgFusion', drop_path_rate=0.0),
               decode_head=dict(type='DAFormerHeadFusion', **HEAD_CFG,
                                decoder_params=_decoder_params(train_type='cs2dsec_image+events_together', share_decoder=True)),
               train_type='cs2dsec_image+events_together', train_cfg=dict(), test_cfg=dict(mode='whole'))
    model = ns.builder.build_segmentor(cfg)
    seeded_fill(model, 101)
    model.train()
    B, H, W = 1, 64, 64
    inputs = {k: seeded_randn((B, 3, H, W), 101, k) for k in ('image', 'events', 'img_self_res')}
    gt = _label(B, H, W, 101)
    fcfg = dict(loss_weight={'image': 0.5, 'events': 0.5, 'fusion': 0.5, 'img_self_res': 0.25}, gradual_rate=0.0)
    losses, pred = model.forward_train(inputs, gt, return_feat=True, cfg=fcfg)
    losses.pop('features')
    losses['decode.loss_seg'].backward()
    save('segmentor_train', loss_seg=losses['decode.loss_seg'], acc_seg=losses['decode.acc_seg'], gt=gt,
         **{k: v for k, v in pred.items()}, **param_grads(model, 96))
    model.eval()
    with torch.no_grad():
        out = model.encode_decode(inputs['image'], inputs['events'], output_features=True, test_cfg=fcfg)
    save('segmentor_teacher', **{k: v for k, v in out.items() if v is not None})
    with open(os.path.join(HERE, 'segmentor_keys.json'), 'w') as f:
        json.dump(sorted(model.state_dict().keys()), f, indent=0)


def dacs_cfg(G_path):
    """reduced-width FusionEncoderDecoder under the reference's DACS (launcher defaults of SURVEY.md appendix A; DropPath / Dropout
    rates 0 so that the only draws are the ones recorded below)"""
    bb = dict(type='MixVisionTransformer', patch_size=4, embed_dims=DACS_DIMS, num_heads=[1, 2, 5, 8], mlp_ratios=[4, 4, 4, 4],
              qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), depths=[1, 1, 1, 1], sr_ratios=[8, 4, 2, 1],
              drop_path_rate=0.0)
    head = dict(type='DAFormerHeadFusion', in_channels=DACS_DIMS, in_index=[0, 1, 2, 3], channels=DACS_CH, dropout_ratio=0.0,
                num_classes=19, norm_cfg=dict(type='BN', requires_grad=True), align_corners=False,
                decoder_params=dict(embed_dims=DACS_CH, embed_cfg=dict(type='mlp', act_cfg=None, norm_cfg=None),
                                    embed_neck_cfg=dict(type='mlp', act_cfg=None, norm_cfg=None),
                                    fusion_cfg=dict(type='aspp', sep=True, dilations=(1, 6, 12, 18), pool=False,
                                                    act_cfg=dict(type='ReLU'), norm_cfg=dict(type='BN', requires_grad=True)),
                                    train_type='cs2dsec_image+events_together', share_decoder=True),
                loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0))
    model = dict(type='FusionEncoderDecoder', pretrained=None, backbone_image=dict(bb), backbone_events=dict(bb),
                 fusion_module=dict(type='AttentionAvgFusion', in_channels=DACS_DIMS, drop_path_rate=0.0), decode_head=head,
                 train_type='cs2dsec_image+events_together', train_cfg=dict(work_dir='/tmp/cmda_golden_dacs'), test_cfg=dict(mode='whole'))
    return dict(model=model, max_iters=40000, alpha=0.999, pseudo_threshold=0.968, pseudo_weight_ignore_top=0,
                pseudo_weight_ignore_bottom=0, imnet_feature_dist_lambda=0, imnet_feature_dist_classes=None,
                imnet_feature_dist_scale_min_ratio=None, mix='class', blur=True, color_jitter_strength=0.2,
                color_jitter_probability=0.2, debug_img_interval=10 ** 9, print_grad_magnitude=False,
                train_type='cs2dsec_image+events_together',
                forward_cfg=dict(loss_weight={'image': 0.5, 'events': 0.5, 'fusion': 0.5, 'img_self_res': 0.25}, gradual_rate=0.0),
                cyclegan_itrd2en_path=G_path, img_self_res_reg='no', mixed_image_to_mixed_isr=True, random_choice_thres='0.5',
                shift_type='random', isr_parms=dict(val_range=[0.01, 1.01], _threshold=0.005, _clip_range=0.1, shift_pixel=1),
                sky_mask=None)


def fingerprint(t, n=24):
    return sample_grad(t, n)


@case
def dacs_step():
    """the reference's OWN DACS.train_step (mmseg/models/uda/dacs.py:274-315, forward_train :357-860,
    _init_ema_weights / _update_ema :250-272) for local_iter 0, 1, 2 on one 512 x 512 pair (the mixing code hard-codes width = height
    = 512, :735), reduced-width model on the CPU, torch AdamW between the iterations; plus _update_ema at it = 1500.
    kornia is absent: `random.uniform` is patched so that the colour-jitter and blur gates stay closed (values recorded); the class
    draw (np.random.choice), the events / ISR choice (torch.rand) are seeded and recorded.  Debug plotting (local_iter % interval == 0
    at iteration 0) is stubbed in the module's namespace."""
    import random
    import tempfile
    nn.Module.cuda = lambda self, *a, **k: self
    mm = sys.modules['mmseg.models']
    for k in ('BaseSegmentor', 'BaseSegmentorEvents', 'BaseSegmentorFusion'):
        setattr(mm, k, getattr(ns.seg_base, k))
    cg = sys.modules['mmseg.models.cyclegan']
    cg.define_G, cg.LightNet = ns.cyclegan.define_G, getattr(ns.cyclegan, 'LightNet', None)
    D = ref_shim.load('mmseg.models.uda.dacs')
    # the generator checkpoint DACS.__init__ loads (:96-103)
    G = ns.cyclegan.define_G()
    seeded_fill(G, 113)
    gpath = os.path.join(tempfile.mkdtemp(), 'G.pth')
    torch.save(G.state_dict(), gpath)
    dacs = D.DACS(**dacs_cfg(gpath))
    seeded_fill(dacs.model, 111)
    seeded_fill(dacs.ema_model, 112)      # iteration 0 must overwrite it
    with torch.no_grad():                 # a peaky classifier: some pixels must clear the 0.968 confidence threshold (:701-705)
        dacs.model.decode_head.conv_seg.weight.mul_(DACS_SEG_SCALE)
    dacs.train()
    # debug plotting off (module namespace, not the source)
    class _NoPlot:
        def __getattr__(self, n):
            if n.startswith('__'):
                raise AttributeError(n)
            if n == 'subplots':
                axs = np.empty((8, 8), dtype=object)
                for i in range(8):
                    for j in range(8):
                        axs[i, j] = _NoPlot()
                return lambda *a, **k: (_NoPlot(), axs)
            return _NoPlot()

        def __call__(self, *a, **k):
            return _NoPlot()
    D.plt, D.subplotimg = _NoPlot(), (lambda *a, **k: None)
    opt = torch.optim.AdamW(dacs.model.parameters(), lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01)
    src, tg = dacs_batch()
    out = {}
    captured = {}
    model = dacs.get_model()
    orig_ft, orig_ed = model.forward_train, dacs.get_ema_model().encode_decode

    def ft(inputs, gt, seg_weight=None, return_feat=False, cfg=None):
        if seg_weight is not None:      # the mixed step (:820-860)
            captured.update(mixed_img=inputs['image'].detach().clone(), mixed_events=inputs['events'].detach().clone(),
                            mixed_isr=inputs['img_self_res'].detach().clone(), mixed_lbl=gt.detach().clone(),
                            mixed_weight=seg_weight.detach().clone())
        return orig_ft(inputs, gt, seg_weight=seg_weight, return_feat=return_feat, cfg=cfg)

    def ed(*a, **k):
        o = orig_ed(*a, **k)
        captured['teacher_fusion'] = o['fusion_output'].detach().clone()
        return o
    model.forward_train, dacs.get_ema_model().encode_decode = ft, ed
    uniform = random.uniform
    gates = [(0.13, 0.31, 0.5), (0.07, 0.44, 0.9), (0.18, 0.12, 0.3)]   # (colour-jitter u <= p = 0.2, blur u <= 0.5, sigma)
    for it in range(3):
        seq = iter(gates[it])
        random.uniform = lambda a, b: next(seq)
        torch.manual_seed(500 + it)
        np.random.seed(500 + it)
        # replay of the class draw get_class_masks will make (dacs_transforms.py:101-112)
        classes = torch.unique(src['label'])
        n = classes.shape[0]
        st = np.random.get_state()
        chosen = [classes[torch.Tensor(np.random.choice(n, int((n + n % 2) / 2), replace=False)).long()] for _ in range(1)]
        np.random.set_state(st)
        batch = dict(source={k: v.clone() for k, v in src.items()}, target={k: v.clone() for k, v in tg.items()})
        res = dacs.train_step(batch, opt)
        random.uniform = uniform
        lv = res['log_vars']
        prob, plabel = torch.max(torch.softmax(captured['teacher_fusion'], dim=1), dim=1)
        out[f'it{it}.losses'] = np.array([lv['decode.loss_seg'], lv['decode.acc_seg'], lv['mix.decode.loss_seg'], lv['mix.decode.acc_seg']])
        out[f'it{it}.choice'] = float(dacs.forward_cfg['isr_events_fusion_choice'])
        out[f'it{it}.gates'] = np.array(gates[it])
        out[f'it{it}.classes'] = chosen[0]
        out[f'it{it}.pseudo_label'] = plabel.to(torch.uint8)
        out[f'it{it}.pseudo_conf'] = (prob >= 0.968).sum()
        out[f'it{it}.teacher_fusion_s'] = captured['teacher_fusion'][..., ::16, ::16]
        out[f'it{it}.mixed_img_s'] = captured['mixed_img'][..., ::4, ::4]
        out[f'it{it}.mixed_events_s'] = captured['mixed_events'][:, :1, ::4, ::4]
        out[f'it{it}.mixed_isr_s'] = captured['mixed_isr'][:, :1, ::2, ::2].half()
        out[f'it{it}.mixed_lbl'] = captured['mixed_lbl'].to(torch.uint8)
        out[f'it{it}.mixed_weight_s'] = captured['mixed_weight'][..., ::8, ::8]
        out[f'it{it}.num_samples'] = res['num_samples']
        for k, p in dacs.model.named_parameters():
            out[f'it{it}.grad.{k}'] = fingerprint(p.grad)
            out[f'it{it}.param.{k}'] = fingerprint(p.data)
        for k, p in dacs.ema_model.named_parameters():
            out[f'it{it}.ema.{k}'] = fingerprint(p.data)
        for k, b_ in dacs.model.named_buffers():
            if 'running' in k and k.startswith('decode_head.fuse_layer_image'):
                out[f'it{it}.bn.{k}'] = b_.clone()
        print('iteration', it, lv, 'choice', out[f'it{it}.choice'], 'classes', chosen[0].tolist(), 'conf', int(out[f'it{it}.pseudo_conf']))
    assert dacs.local_iter == 3
    dacs._update_ema(1500)
    for k, p in dacs.ema_model.named_parameters():
        out[f'ema1500.{k}'] = fingerprint(p.data)
    save('dacs_step', **out)


if __name__ == '__main__':
    names = sys.argv[1:] or list(CASES)
    for n in names:
        print('==', n)
        CASES[n]()
