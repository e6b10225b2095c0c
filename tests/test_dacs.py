"""One full DACS iteration (source step, EMA teacher pseudo-labels, ClassMix + on-device ISR, mixed step) on the HIP
kernels against the same iteration composed from the oracle's pieces (reduced-depth MiT so it runs in the emulator)."""
import os
import random
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
from weights import seeded_fill, seeded_randn  # noqa: E402

import cmda_amd  # noqa: E402,F401
import cmda_amd.runtime as rt  # noqa: E402
from cmda_amd.registry import build_train_model  # noqa: E402
from conftest import assert_close  # noqa: E402
from oracle import fusion as ofu, head as ohd, mit as omit, segmentor as oseg, uda as ouda  # noqa: E402

DEPTHS = [1, 1, 1, 1]
DIMS = [32, 64, 160, 256]   # reduced widths (head dim 32) keep the emulator run short; the full MiT-B5 runs in test_modules
CH = 64
ISR = dict(val_range=[0.01, 1.01], _threshold=0.005, _clip_range=0.1, shift_pixel=1)
FCFG = dict(loss_weight={'image': 0.5, 'events': 0.5, 'fusion': 0.5, 'img_self_res': 0.25}, gradual_rate=0.0)


def make_cfg():
    bb = dict(type='MixVisionTransformer', embed_dims=DIMS, num_heads=[1, 2, 5, 8], qkv_bias=True,
              depths=DEPTHS, sr_ratios=[8, 4, 2, 1], drop_path_rate=0.0,
              norm_layer=__import__('functools').partial(torch.nn.LayerNorm, eps=1e-6))
    head = dict(type='DAFormerHeadFusion', in_channels=DIMS, in_index=[0, 1, 2, 3], channels=CH,
                dropout_ratio=0.0, num_classes=19, norm_cfg=dict(type='BN', requires_grad=True), align_corners=False,
                decoder_params=dict(embed_dims=CH, embed_cfg=dict(type='mlp', act_cfg=None, norm_cfg=None),
                                    embed_neck_cfg=dict(type='mlp', act_cfg=None, norm_cfg=None),
                                    fusion_cfg=dict(type='aspp', sep=True, dilations=(1, 6, 12, 18), pool=False,
                                                    act_cfg=dict(type='ReLU'), norm_cfg=dict(type='BN', requires_grad=True)),
                                    train_type='cs2dsec_image+events_together', share_decoder=True),
                loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0))
    model = dict(type='FusionEncoderDecoder', backbone_image=dict(bb), backbone_events=dict(bb),
                 fusion_module=dict(type='AttentionAvgFusion', in_channels=DIMS, drop_path_rate=0.0), decode_head=head,
                 train_type='cs2dsec_image+events_together', train_cfg=dict(), test_cfg=dict(mode='whole'))
    uda = dict(type='DACS', alpha=0.999, pseudo_threshold=0.968, pseudo_weight_ignore_top=0, pseudo_weight_ignore_bottom=0,
               imnet_feature_dist_lambda=0, imnet_feature_dist_classes=None, imnet_feature_dist_scale_min_ratio=None,
               mix='class', blur=False, color_jitter_strength=0.2, color_jitter_probability=2.0, debug_img_interval=1000,
               print_grad_magnitude=False, train_type='cs2dsec_image+events_together', forward_cfg=FCFG,
               cyclegan_itrd2en_path='', img_self_res_reg='no', mixed_image_to_mixed_isr=True, random_choice_thres='0.5',
               shift_type='rightdown', isr_parms=ISR, sky_mask=None)
    return dict(model=model, uda=uda, runner=dict(type='IterBasedRunner', max_iters=40000))


def oracle_student():
    return oseg.FusionEncoderDecoder(backbone_image=omit.MixVisionTransformer(embed_dims=DIMS, depths=DEPTHS, drop_path_rate=0.0),
                                     backbone_events=omit.MixVisionTransformer(embed_dims=DIMS, depths=DEPTHS, drop_path_rate=0.0),
                                     fusion_module=ofu.AttentionAvgFusion(in_channels=DIMS, drop_path_rate=0.0),
                                     decode_head=ohd.DAFormerHeadFusion(in_channels=DIMS, channels=CH, embed_dims=CH,
                                                                        dropout_ratio=0.0, share_decoder=True))


def test_dacs_iteration_matches_oracle(tgt):
    rt.set_compute_dtype(torch.float32)
    B, H, W = 2, 64, 64
    dacs = build_train_model(make_cfg())
    seeded_fill(dacs.model, 7)
    seeded_fill(dacs.ema_model, 8)  # different from the student: iteration 0 must overwrite it
    dacs.to(tgt.device).train()
    g = torch.Generator().manual_seed(3)
    lab = torch.randint(0, 6, (B, 1, H // 8, W // 8), generator=g).repeat_interleave(8, 2).repeat_interleave(8, 3)
    lab[0, 0, :4] = 255
    src = dict(image=seeded_randn((B, 3, H, W), 7, 'img'), img_time_res=seeded_randn((B, 3, H, W), 7, 'itr'),
               img_self_res=seeded_randn((B, 3, H, W), 7, 'isr').clamp(-1, 1), label=lab)
    tg = dict(warp_image=seeded_randn((B, 3, H, W), 7, 'nimg'), events_vg=seeded_randn((B, 3, H, W), 7, 'nev').clamp(-1, 1),
              warp_img_self_res=seeded_randn((B, 3, H, W), 7, 'nisr').clamp(-1, 1))
    batch = dict(source={k: tgt.to(v) for k, v in src.items()}, target={k: tgt.to(v) for k, v in tg.items()})

    # ---- oracle iteration --------------------------------------------------------------------------------------------
    ref = oracle_student()
    seeded_fill(ref, 7).train()
    ema = oracle_student()
    seeded_fill(ema, 8).train()
    torch.manual_seed(11), random.seed(11), np.random.seed(11)
    choice = torch.rand(1)
    ouda.update_ema(list(ema.parameters()), list(ref.parameters()), 0, 0.999)
    _cj = random.uniform(0, 1)
    use_events = bool(choice > 0.5)
    inputs = {'image': src['image'], 'events': src['img_time_res'], 'img_self_res': src['img_self_res']}
    l_s, _ = ref.forward_train(inputs, lab, return_feat=False, cfg=FCFG)
    l_s['decode.loss_seg'].backward()
    with torch.no_grad():
        out = ema.encode_decode(tg['warp_image'], tg['events_vg'] if use_events else tg['warp_img_self_res'],
                                output_features=True, test_cfg=FCFG)
        prob, plabel = torch.softmax(out['fusion_output'], dim=1).max(dim=1)
        pw = (prob.ge(0.968).sum().item() / plabel.numel()) * torch.ones(prob.shape)
    chosen = ouda.choose_classes(lab, np.random)
    mixed_img, mixed_ev, mixed_lbl, mixed_w, mixed_isr = [], [], [], [], []
    for i in range(B):
        m = ouda.class_mask(lab[i], chosen[i])
        mixed_img.append(ouda.one_mix(m, src['image'][i], tg['warp_image'][i])[None])
        mixed_ev.append(ouda.one_mix(m, src['img_time_res'][i], tg['events_vg'][i])[None])
        mixed_lbl.append(ouda.one_mix(m, lab[i][0], plabel[i])[None])
        mixed_w.append(ouda.one_mix(m, torch.ones(H, W), pw[i]))
        mixed_isr.append(ouda.mixed_image_to_isr(mixed_img[-1], 1, [0.01, 1.01], 0.005, 0.1, 'rightdown'))
    mixed_img, mixed_ev, mixed_lbl = torch.cat(mixed_img), torch.cat(mixed_ev), torch.cat(mixed_lbl)
    mixed_w, mixed_isr = torch.cat(mixed_w), torch.cat(mixed_isr)
    l_m, _ = ref.forward_train({'image': mixed_img, 'events': mixed_ev, 'img_self_res': mixed_isr}, mixed_lbl,
                               seg_weight=mixed_w, cfg=FCFG)
    l_m['decode.loss_seg'].backward()

    # ---- HIP iteration (same RNG streams) --------------------------------------------------------------------------------
    torch.manual_seed(11), random.seed(11), np.random.seed(11)
    log_vars = dacs(**batch)
    mix = dacs.last_mix
    assert torch.equal(mix['classes'].cpu()[0][mix['classes'].cpu()[0] >= 0], chosen[0])
    agree = (mix['pseudo_label'].cpu() == plabel).float().mean().item()
    assert agree > 0.999, f'pseudo-label agreement {agree}'
    assert_close(mix['mixed_img'], mixed_img, 0, name='mixed image')
    same = (mix['mixed_lbl'].cpu() == mixed_lbl).float().mean().item()
    assert same > 0.999
    assert_close(mix['mixed_isr'], mixed_isr, 1e-5, atol=1e-6, name='mixed ISR')
    assert_close(mix['pseudo_weight'], mixed_w, 1e-4, name='mixed weight')
    assert_close(log_vars['decode.loss_seg'], l_s['decode.loss_seg'], 1e-4, name='source loss')
    assert_close(log_vars['mix.decode.loss_seg'], l_m['decode.loss_seg'], 2e-3, name='mix loss')
    for (n1, p), (n2, q) in zip(dacs.ema_model.named_parameters(), ema.named_parameters()):
        assert_close(p.data, q.data, 0, name='ema ' + n1)
    worst = 0.0
    for (n1, p), (n2, q) in zip(dacs.model.named_parameters(), ref.named_parameters()):
        assert n1 == n2
        e = (p.grad.cpu() - q.grad).abs().max().item() / (q.grad.abs().max().item() + 1e-12)
        worst = max(worst, e)
    assert worst < 5e-2, f'worst accumulated-gradient relative error {worst}'
