"""One full DACS iteration (Motion-Extractor generator, EMA teacher, source step, pseudo-labels, ClassMix + colour jitter +
blur + on-device ISR, mixed step) on the HIP kernels against the same iteration of oracle/dacs_iter.py (which follows
mmseg/models/uda/dacs.py:357-860).  The host decisions (events/ISR choice, jitter / blur gates and parameters, class draw)
are taken from the HIP run (`DACS.last_draws`) and injected into the oracle, so both sides compute the same iteration.

  * reduced width / depth, fp32: runs in the CPU emulator and on the GPU;
  * MiT-B5 widths (head dim 64 -> the fused attention kernel in bf16), depth 1 per stage, fp32 and bf16: GPU only;
  * the same iteration replayed as ONE hipGraph must reproduce the eager result: GPU only.
"""
import functools
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
from conftest import assert_close, assert_close_robust, check_ge, check_le  # noqa: E402
from oracle import cyclegan as ocg, dacs_iter, fusion as ofu, head as ohd, mit as omit, segmentor as oseg  # noqa: E402

DEPTHS = [1, 1, 1, 1]
# bf16 generator output against the fp32 oracle, of its range (~1): operand rounding over 24 conv + InstanceNorm layers -- the CPU
# model of the rounding points (tools/dbg/gen_bf16_model.py) gives 3.9e-2 max / 2.9e-2 at the 99.9th percentile at 128 x 128 for
# fp32 conv outputs + fp32 residual stream (round 3, bf16 everywhere: 5.1e-2 / 3.8e-2); only split-bf16 operands remove it (7e-5).
# The max over 32 k pixels depends on the atomic summation order of the InstanceNorm statistics, so the gate is the 99.9th
# percentile with a loose hard max (VERDICT r03 #1b).
GEN_BF16_P999, GEN_BF16_MAX = 6.5e-2, 0.12   # (3.0e-2 / 4.8e-2 worst of the audited runs: profiles/r05_test_margins.txt)
SMALL = dict(dims=[32, 64, 160, 256], ch=64)   # reduced widths (head dim 32) keep the emulator run short
FULLW = dict(dims=[64, 128, 320, 512], ch=256)  # MiT-B5 widths: head dim 64, the bench's kernel selection
ISR = dict(val_range=[0.01, 1.01], _threshold=0.005, _clip_range=0.1, shift_pixel=1)
FCFG = dict(loss_weight={'image': 0.5, 'events': 0.5, 'fusion': 0.5, 'img_self_res': 0.25}, gradual_rate=0.0)


def make_cfg(dims, ch, generator=True, blur=True, jitter_p=0.2, shift_type='random', train_type='cs2dsec_image+events_together',
             fusion='AttentionAvgFusion', ignore_top=0, ignore_bottom=0, depths=None, drop_path_rate=0.0, fusion_drop_path=0.0,
             dropout_ratio=0.0):
    bb = dict(type='MixVisionTransformer', embed_dims=dims, num_heads=[1, 2, 5, 8], qkv_bias=True,
              depths=depths or DEPTHS, sr_ratios=[8, 4, 2, 1], drop_path_rate=drop_path_rate,
              norm_layer=functools.partial(torch.nn.LayerNorm, eps=1e-6))
    head = dict(type='DAFormerHeadFusion', in_channels=dims, in_index=[0, 1, 2, 3], channels=ch,
                dropout_ratio=dropout_ratio, num_classes=19, norm_cfg=dict(type='BN', requires_grad=True), align_corners=False,
                decoder_params=dict(embed_dims=ch, embed_cfg=dict(type='mlp', act_cfg=None, norm_cfg=None),
                                    embed_neck_cfg=dict(type='mlp', act_cfg=None, norm_cfg=None),
                                    fusion_cfg=dict(type='aspp', sep=True, dilations=(1, 6, 12, 18), pool=False,
                                                    act_cfg=dict(type='ReLU'), norm_cfg=dict(type='BN', requires_grad=True)),
                                    train_type=train_type, share_decoder=True),
                loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0))
    model = dict(type='FusionEncoderDecoder', backbone_image=dict(bb), backbone_events=dict(bb),
                 fusion_module=dict(type=fusion, in_channels=dims, drop_path_rate=fusion_drop_path), decode_head=head,
                 train_type=train_type, train_cfg=dict(), test_cfg=dict(mode='whole'))
    uda = dict(type='DACS', alpha=0.999, pseudo_threshold=0.968, pseudo_weight_ignore_top=ignore_top,
               pseudo_weight_ignore_bottom=ignore_bottom,
               imnet_feature_dist_lambda=0, imnet_feature_dist_classes=None, imnet_feature_dist_scale_min_ratio=None,
               mix='class', blur=blur, color_jitter_strength=0.2, color_jitter_probability=jitter_p, debug_img_interval=1000,
               print_grad_magnitude=False, train_type=train_type, forward_cfg=dict(FCFG),
               cyclegan_itrd2en_path='random' if generator else '', img_self_res_reg='no', mixed_image_to_mixed_isr=True,
               random_choice_thres='0.5', shift_type=shift_type, isr_parms=dict(ISR), sky_mask=None)
    return dict(model=model, uda=uda, runner=dict(type='IterBasedRunner', max_iters=40000))


def oracle_student(dims, ch, fusion='AttentionAvgFusion', depths=None, drop_path_rate=0.0, fusion_drop_path=0.0, dropout_ratio=0.0):
    depths = depths or DEPTHS
    return oseg.FusionEncoderDecoder(backbone_image=omit.MixVisionTransformer(embed_dims=dims, depths=depths, drop_path_rate=drop_path_rate),
                                     backbone_events=omit.MixVisionTransformer(embed_dims=dims, depths=depths, drop_path_rate=drop_path_rate),
                                     fusion_module=getattr(ofu, fusion)(in_channels=dims, drop_path_rate=fusion_drop_path),
                                     decode_head=ohd.DAFormerHeadFusion(in_channels=dims, channels=ch, embed_dims=ch,
                                                                        dropout_ratio=dropout_ratio, share_decoder=True))


def make_batch(B, H, W):
    g = torch.Generator().manual_seed(3)
    lab = torch.randint(0, 6, (B, 1, H // 8, W // 8), generator=g).repeat_interleave(8, 2).repeat_interleave(8, 3)
    lab[0, 0, :4] = 255
    itr = seeded_randn((B, 1, H, W), 7, 'itr').clamp(-1, 1).repeat(1, 3, 1, 1)
    src = dict(image=seeded_randn((B, 3, H, W), 7, 'img'), img_time_res=itr,
               img_self_res=seeded_randn((B, 3, H, W), 7, 'isr').clamp(-1, 1), label=lab)
    tg = dict(warp_image=seeded_randn((B, 3, H, W), 7, 'nimg'), events_vg=seeded_randn((B, 3, H, W), 7, 'nev').clamp(-1, 1),
              warp_img_self_res=seeded_randn((B, 3, H, W), 7, 'nisr').clamp(-1, 1))
    return src, tg


def oracle_draws(d):
    """DACS.last_draws -> the `draws` argument of oracle.dacs_iter.dacs_iteration (class rows without the -1 padding)"""
    out = {k: d[k] for k in ('choice', 'color_jitter', 'blur', 'sigma', 'jitter')}
    out['classes'] = [row[row >= 0] for row in d['classes']]
    return out


def run_case(tgt, dims, ch, dtype, B=2, H=64, W=64, iters=1, graph=False, shift_type='random', lanes=None, **variant):
    """variant: train_type / fusion / ignore_top / ignore_bottom / generator of the second reference config (cs2dz_image+raw-isr)"""
    rt.set_compute_dtype(dtype)
    dacs = build_train_model(make_cfg(dims, ch, shift_type=shift_type, **variant))
    seeded_fill(dacs.model, 7)
    seeded_fill(dacs.ema_model, 8)  # different from the student: iteration 0 must overwrite it
    if dacs.cyclegan_itrd2en is not None:
        seeded_fill(dacs.cyclegan_itrd2en, 9)
    dacs.to(tgt.device).train()
    src, tg = make_batch(B, H, W)
    batch = dict(source={k: tgt.to(v) for k, v in src.items()}, target={k: tgt.to(v) for k, v in tg.items()})
    fus = variant.get('fusion', 'AttentionAvgFusion')
    ref, ema, G = oracle_student(dims, ch, fus), oracle_student(dims, ch, fus), ocg.ResnetGenerator().eval()
    seeded_fill(ref, 7).train()
    seeded_fill(ema, 8).train()
    seeded_fill(G, 9)
    torch.manual_seed(11), random.seed(11), np.random.seed(11)
    if graph:
        dacs.enable_graph(warmup_iters=1)
        if lanes is not None:
            dacs.graph_lane_set = set(lanes)
    outs = []
    for it in range(iters):
        for p in dacs.model.parameters():
            if p.grad is not None:
                p.grad.zero_()
        log_vars = dacs(**batch)
        mix = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in dacs.last_mix.items()}
        grads = {n: p.grad.detach().cpu().clone() for n, p in dacs.model.named_parameters()}
        for p in ref.parameters():
            p.grad = None
        o = dacs_iter.dacs_iteration(ref, ema, G if dacs.cyclegan_itrd2en is not None else None, src, tg, local_iter=it,
                                     forward_cfg=FCFG, isr_parms=ISR, shift_type=shift_type, draws=oracle_draws(dacs.last_draws),
                                     train_type=variant.get('train_type', 'cs2dsec_image+events_together'),
                                     ignore_top=variant.get('ignore_top', 0), ignore_bottom=variant.get('ignore_bottom', 0))
        outs.append(({k: v.detach().clone() for k, v in log_vars.items()}, mix, grads, o,
                     {n: q.grad.clone() for n, q in ref.named_parameters()}))
    # BatchNorm running statistics of the student after the iterations: source step first, then the mixed step, branch by branch
    mine = dict(dacs.model.named_buffers())
    seen = 0
    for n, b in ref.named_buffers():
        if 'running_' in n:
            assert_close(mine[n], b, 1e-4 if dtype == torch.float32 else 5e-2, atol=1e-5, name='student ' + n)
            seen += 1
    assert seen > 0
    return dacs, ema, outs


def check_iteration(out, exact, tol_loss, tol_grad, label_agree=0.999, tol_gen=None):
    """tol_gen: bound on the generator output (of its range ~1); default 2e-4 in the exact-fp32 mode (9e-6 measured), 5e-4 is used for
    the split-bf16 mode (7e-5 measured: 24 convolution + InstanceNorm layers)"""
    log_vars, mix, grads, o, ref_grads = out
    if o['day_events'] is not None:
        if exact:
            assert_close(mix['day_events'], o['day_events'], tol_gen or 2e-4, name='generator output (day events)')
        else:
            assert_close_robust(mix['day_events'], o['day_events'], GEN_BF16_P999, GEN_BF16_MAX, name='generator output (day events)')
    agree = (mix['pseudo_label'].cpu() == o['pseudo_label']).float().mean().item()
    check_ge('pseudo-label agreement', agree, label_agree, strict=True)
    if exact:
        assert_close(mix['mixed_img'], o['mixed_img'], 1e-4, atol=2e-4, name='mixed image', outlier_frac=1e-3, outlier_rtol=2.0)
        same = (mix['mixed_lbl'].cpu() == o['mixed_lbl']).float().mean().item()
        check_ge('mixed-label agreement', same, label_agree, strict=True)
        # uint8 truncation of the jittered image can move one gray level where the two colour-jitter implementations differ
        # in the last bit: the ISR is compared on the bulk
        assert_close(mix['mixed_isr'], o['mixed_isr'], 1e-5, atol=1e-6, name='mixed ISR', outlier_frac=5e-3, outlier_rtol=2.0)
        assert_close(mix['pseudo_weight'], o['mixed_weight'], 1e-3, name='mixed weight')
    assert_close(log_vars['decode.loss_seg'], o['decode.loss_seg'], tol_loss, name='source loss')
    assert_close(log_vars['mix.decode.loss_seg'], o['mix.decode.loss_seg'], tol_loss * 20, name='mix loss')
    worst = 0.0
    for n, q in ref_grads.items():
        e = (grads[n] - q).abs().max().item() / (q.abs().max().item() + 1e-12)
        worst = max(worst, e)
    check_le('worst accumulated-gradient relative error', worst, tol_grad, strict=True)


def test_dacs_iteration_matches_oracle(tgt):
    """reduced widths, fp32, generator + jitter + blur + random ISR direction ON"""
    dacs, ema, outs = run_case(tgt, SMALL['dims'], SMALL['ch'], torch.float32)
    d = dacs.last_draws
    assert d['jitter'] is not None and len(d['jitter']) == 2 and d['jitter'][0] != d['jitter'][1], 'per-sample jitter draws'
    check_iteration(outs[0], True, 1e-4, 5e-2)
    for (n1, p), (n2, q) in zip(dacs.ema_model.named_parameters(), ema.named_parameters()):
        assert_close(p.data, q.data, 0, name='ema ' + n1)


def test_dacs_iteration_x3_matches_oracle(tgt):
    """the tolerance-meeting mode: fp32 storage, every GEMM on split-bf16 operands (runtime.set_gemm_x3, csrc/gemm_x3.hip) -- the
    fp32 bounds of test_dacs_iteration_matches_oracle hold unchanged"""
    rt.set_gemm_x3(True)
    try:
        dacs, ema, outs = run_case(tgt, SMALL['dims'], SMALL['ch'], torch.float32)
        check_iteration(outs[0], True, 1e-4, 0.1, tol_gen=5e-4)   # (exact fp32: worst gradient 6e-3, generator 9e-6; split-bf16: 2e-2, 7e-5)
    finally:
        rt.set_gemm_x3(False)


def test_dacs_iteration_second_config_matches_oracle(tgt):
    """configs/fusion/cs2dz_image+raw-isr_b5.py: AttentionFusion (concat) instead of the averaging fusion, no
    events and no generator -- the event encoder sees the ISR --, three decoder branches, the pseudo-weight's top / bottom rows
    zeroed (dacs.py:707-710); the student's source + mixed steps still run as one pass (6 BatchNorm groups)"""
    dacs, ema, outs = run_case(tgt, SMALL['dims'], SMALL['ch'], torch.float32, train_type='cs2dz_image+raw-isr',
                               fusion='AttentionFusion', generator=False, ignore_top=3, ignore_bottom=9)
    assert dacs.cyclegan_itrd2en is None
    w = outs[0][1]['pseudo_weight']
    assert w.shape[-2:] == (64, 64)
    check_iteration(outs[0], True, 1e-4, 5e-2)
    for (n1, p), (n2, q) in zip(dacs.ema_model.named_parameters(), ema.named_parameters()):
        assert_close(p.data, q.data, 0, name='ema ' + n1)


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['f32', 'x3', 'bf16'])
def test_dacs_iteration_full_width_gpu(mode):
    """MiT-B5 widths (head dim 64): in bf16 this is the bench's kernel selection (fused attention, bf16 MFMA GEMMs)"""
    from conftest import Target
    from cmda_amd import _lib
    _lib._unbind_for_tests()
    if not torch.cuda.is_available():
        pytest.skip('no GPU on this machine')
    tgt = Target('gpu')
    dt = torch.bfloat16 if mode == 'bf16' else torch.float32
    rt.set_gemm_x3(mode == 'x3')
    try:
        dacs, ema, outs = run_case(tgt, FULLW['dims'], FULLW['ch'], dt, H=128, W=128)
        if mode != 'bf16':
            check_iteration(outs[0], True, 1e-4, 5e-2 if mode == 'f32' else 0.1, tol_gen=None if mode == 'f32' else 5e-4)
        else:
            check_iteration(outs[0], False, 2e-2, 0.55, label_agree=0.96)   # (worst gradient 0.19-0.27, labels 0.9876 measured, eight runs: 2x)
    finally:
        rt.set_gemm_x3(False)
        rt.set_compute_dtype(torch.float32)


@pytest.mark.gpu
@pytest.mark.parametrize('lanes', [None, ('enc', 'wgrad'), (), ('enc', 'T'), ('enc', 'T', 'Tenc', 'wq')])
def test_dacs_graph_replay_matches_oracle(lanes):
    """iteration 0 eager (warm-up), iterations 1-2 captured / replayed as hipGraph segments: each must still match the oracle's
    iteration with the same draws (different draws per iteration: the gates and parameters travel through the control block).
    lanes: the default side lane; + the block-level weight-gradient lane (runtime.lane_batch); no side lane at all."""
    from conftest import Target
    from cmda_amd import _lib
    _lib._unbind_for_tests()
    if not torch.cuda.is_available():
        pytest.skip('no GPU on this machine')
    tgt = Target('gpu')
    dacs, ema, outs = run_case(tgt, SMALL['dims'], SMALL['ch'], torch.float32, iters=3, graph=True, lanes=lanes)
    assert dacs._graph is not None, 'the iteration was not captured'
    for it, out in enumerate(outs):
        print(f'iteration {it}: source loss {out[0]["decode.loss_seg"].item():.6f} vs {out[3]["decode.loss_seg"].item():.6f}, mix loss '
              f'{out[0]["mix.decode.loss_seg"].item():.6f} vs {out[3]["mix.decode.loss_seg"].item():.6f}; draws {dacs.last_draws["choice"]:.3f}')
    for it, out in enumerate(outs):
        check_iteration(out, True, 1e-4, 5e-2)
    for (n1, p), (n2, q) in zip(dacs.ema_model.named_parameters(), ema.named_parameters()):
        assert_close(p.data, q.data, 1e-6, name='ema ' + n1)


def _upsampled(logits_nhwc, H, W):
    from cmda_amd import ops
    return ops.upsample_logits_nchw(logits_nhwc.contiguous(), H, W).cpu()


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['f32', 'x3', 'bf16'])
def test_dacs_iteration_full_depth_512_gpu(mode):
    """THE BENCH'S CONFIGURATION against the oracle (VERDICT r02 #4a): full-depth MiT-B5 encoders (3, 6, 40, 3), 512 x 512, 2 source
    + 2 target samples, generator + colour jitter + blur + random ISR direction ON, DropPath / Dropout2d OFF (the draws that remain
    are injected from the HIP run).  fp32: source / mixed loss 1e-4, teacher logits 1e-3 of their range, pseudo-labels equal up to
    numerical ties (>= 99.99 %), mixed inputs equal; bf16 (the speed mode the bench line runs): reported and bounded.
    CMDA_TEST_FULL_B: samples per domain (default 2; the CPU oracle holds every activation of 3 + 3 encoder passes)."""
    import time
    from conftest import Target
    from cmda_amd import _lib
    _lib._unbind_for_tests()
    if not torch.cuda.is_available():
        pytest.skip('no GPU on this machine')
    tgt = Target('gpu')
    depths = [3, 6, 40, 3]
    B, S = int(os.environ.get('CMDA_TEST_FULL_B', 2)), 512
    dt = torch.bfloat16 if mode == 'bf16' else torch.float32
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    rt.set_compute_dtype(dt)
    rt.set_gemm_x3(mode == 'x3')   # fp32 storage, split-bf16 GEMMs: the fp32 bounds below hold unchanged
    try:
        dacs = build_train_model(make_cfg(FULLW['dims'], FULLW['ch'], depths=depths))
        seeded_fill(dacs.model, 7)
        seeded_fill(dacs.ema_model, 8)
        seeded_fill(dacs.cyclegan_itrd2en, 9)
        dacs.to(tgt.device).train()
        src, tg = make_batch(B, S, S)
        batch = dict(source={k: tgt.to(v) for k, v in src.items()}, target={k: tgt.to(v) for k, v in tg.items()})
        torch.manual_seed(11), random.seed(11), np.random.seed(11)
        log_vars = dacs(**batch)
        torch.cuda.synchronize()
        mix = dacs.last_mix
        ref, ema, G = (oracle_student(FULLW['dims'], FULLW['ch'], depths=depths), oracle_student(FULLW['dims'], FULLW['ch'], depths=depths),
                       ocg.ResnetGenerator().eval())
        seeded_fill(ref, 7).train()
        seeded_fill(ema, 8).train()
        seeded_fill(G, 9)
        t0 = time.time()
        o = dacs_iter.dacs_iteration(ref, ema, G, src, tg, local_iter=0, forward_cfg=FCFG, isr_parms=ISR, shift_type='random',
                                     draws=oracle_draws(dacs.last_draws))
        t_oracle = time.time() - t0
        # teacher logits (fusion branch), at full resolution, relative to their range
        tl = _upsampled(mix['teacher_logits']['fusion_output'], S, S)
        rl = o['teacher_logits']['fusion_output']
        logit_err = ((tl - rl).abs().max() / rl.abs().max()).item()
        # the same error ELEMENT-WISE: |a - b| / max(|b|, floor) with the floor at 1e-3 of the logit range (a logit near zero has no
        # relative precision to speak of); max and 99.9th percentile.  Every other "rel err" of the suite is the max-norm error
        # relative to the reference tensor's largest element (conftest.assert_close) -- VERDICT r04 weak #4.
        ew = ((tl - rl).abs() / rl.abs().clamp_min(1e-3 * rl.abs().max())).flatten()
        logit_ew_max, logit_ew_p999 = ew.max().item(), ew.kthvalue(max(1, int(0.999 * ew.numel()))).values.item()
        agree = (mix['pseudo_label'].cpu() == o['pseudo_label']).float().mean().item()
        lbl_same = (mix['mixed_lbl'].cpu() == o['mixed_lbl']).float().mean().item()
        ls, lm = log_vars['decode.loss_seg'].item(), log_vars['mix.decode.loss_seg'].item()
        rs, rm = o['decode.loss_seg'].item(), o['mix.decode.loss_seg'].item()
        worst, errs = 0.0, []
        grads = {n: p.grad.detach().cpu() for n, p in dacs.model.named_parameters()}
        for n, q in ref.named_parameters():
            e = (grads[n] - q.grad).abs().max().item() / (q.grad.abs().max().item() + 1e-12)
            errs.append(e)
        errs.sort()

        def _err(a, b):   # (max, 99.9th percentile) of |a - b| relative to max |b|
            d = (a.detach().float().cpu() - b.detach().float().cpu()).abs().flatten()
            sc = max(b.abs().max().item(), 1e-30)
            return d.max().item() / sc, d.kthvalue(max(1, int(0.999 * d.numel()))).values.item() / sc
        gen_max, gen_p999 = _err(mix['day_events'], o['day_events'])          # Motion-Extractor generator output at the bench's size
        mev_max, mev_p999 = _err(mix['mixed_events'], o['mixed_events'])      # ... and the ClassMix'd event tensor the student consumes
        print(f'[{mode}] generator output (day events) rel err max {gen_max:.2e} / 99.9th pct {gen_p999:.2e}; mixed events max {mev_max:.2e} '
              f'/ 99.9th pct {mev_p999:.2e}')
        print(f'[{mode}] teacher logits element-wise rel err (floor 1e-3 of range): max {logit_ew_max:.2e}, 99.9th pct {logit_ew_p999:.2e}')
        print(f'[{mode}] full-depth 512x512 B={B}+{B}: oracle {t_oracle:.0f} s; teacher logits rel err {logit_err:.2e}; pseudo-label '
              f'agreement {agree:.6f}; mixed-label agreement {lbl_same:.6f}; source loss {ls:.6f} vs {rs:.6f}; mixed loss {lm:.6f} vs '
              f'{rm:.6f}; gradient rel err median {errs[len(errs) // 2]:.2e}, 90th pct {errs[int(len(errs) * 0.9)]:.2e}, worst {errs[-1]:.2e}')
        if os.environ.get('CMDA_PARITY_JSON'):   # tools/gpu/parity.sh: the record bench.py's `accuracy` block quotes (profiles/rNN_parity.json)
            import json
            pj = os.environ['CMDA_PARITY_JSON']
            rec = json.load(open(pj)) if os.path.exists(pj) else {}
            rec[{'f32': 'f32', 'x3': 'f32x3', 'bf16': 'bf16'}[mode]] = dict(
                logit_err_range=logit_err, logit_err_elementwise_p999=logit_ew_p999, logit_err_elementwise_max=logit_ew_max,
                pseudo_label_agreement=agree, mixed_label_agreement=lbl_same, source_loss=[ls, rs], mixed_loss=[lm, rm],
                gradient_rel_err_median=errs[len(errs) // 2], gradient_rel_err_p90=errs[int(len(errs) * 0.9)], gradient_rel_err_worst=errs[-1],
                generator_err_max=gen_max, generator_err_p999=gen_p999)
            rec['_what'] = (f'HIP DACS iteration against oracle/dacs_iter.py at the bench configuration: full-depth MiT-B5 (3, 6, 40, 3), '
                            f'512 x 512, {B} + {B} samples (tests/test_dacs.py::test_dacs_iteration_full_depth_512_gpu); logit errors of the '
                            f'teacher fusion logits at full resolution: range-relative max, element-wise with a floor at 1e-3 of the range')
            with open(pj, 'w') as fh:
                json.dump(rec, fh, indent=1, sort_keys=True)
        if mode != 'bf16':
            check_le('generator output rel err (512 x 512)', gen_max, 5e-4 if mode == 'x3' else 2e-4)
            check_le('teacher logits rel err', logit_err, 1e-3, strict=True)
            check_le('teacher logits element-wise rel err, 99.9th pct (floor 1e-3 of range)', logit_ew_p999, 2e-2 if mode == 'x3' else 1e-2)   # (6.9e-3 / 8.1e-4 measured)
            check_ge('pseudo-label agreement', agree, 0.9998)   # (numerical ties: 0.99995 measured in both fp32-storage modes)
            check_ge('mixed-label agreement', lbl_same, 0.9998)
            check_le('source loss abs err', abs(ls - rs), 1e-4 * max(1.0, abs(rs)), strict=True)
            check_le('mixed loss abs err', abs(lm - rm), 2e-3 * max(1.0, abs(rm)), strict=True)
            assert_close(mix['mixed_img'], o['mixed_img'], 1e-4, atol=2e-4, name='mixed image', outlier_frac=1e-3, outlier_rtol=2.0)
            check_le('90th-percentile accumulated-gradient rel err', errs[int(len(errs) * 0.9)], 2e-2, strict=True)
        else:
            check_le('bf16 generator output 99.9th pct rel err (512 x 512)', gen_p999, GEN_BF16_P999)
            check_le('bf16 generator output max rel err (512 x 512)', gen_max, GEN_BF16_MAX)
            check_le('bf16 teacher logits rel err', logit_err, 6e-2, strict=True)
            check_ge('bf16 pseudo-label agreement', agree, float(os.environ.get('CMDA_TEST_BF16_LABEL_AGREE', 0.945)), strict=True)
            check_le('bf16 source loss abs err', abs(ls - rs), 2e-2 * max(1.0, abs(rs)), strict=True)
            check_le('bf16 mixed loss abs err', abs(lm - rm), 0.1 * max(1.0, abs(rm)), strict=True)
    finally:
        rt.set_gemm_x3(False)
        rt.set_compute_dtype(torch.float32)


# bounds of test_dacs_train_step_bf16_against_reference_fixture_gpu: about twice the measured distance of the bf16 mode to the reference's own
# step at iteration 0 (profiles/r06_reference_step_fixture_bf16.txt)
# measured: losses 1.8e-4, accuracies 0.022, pseudo-labels 0.98699, confident-pixel count 31, mixed labels 0.99340, mixed events 3.7e-2,
# gradient fingerprints 90th percentile 8.7e-2 / worst 0.276 (the worst tensors are the round-off-level key biases); worst over the three
# fresh-box runs of the final commit (profiles/r06_test_margins.txt): 90th percentile 0.103, worst 0.325, mixed labels 0.9934 -- the
# bounds keep 2x of those
BF16_FIXTURE = dict(loss=1e-3, acc=0.1, labels=0.974, conf=200, mixed_labels=0.985, events=8e-2, grad_p90=0.25, grad_worst=0.8)


@pytest.mark.gpu
def test_dacs_train_step_bf16_against_reference_fixture_gpu():
    """the SPEED mode (bf16 storage and MFMA operands: what the bench line's headline number runs) against the reference's own
    `DACS.train_step` (tests/golden/dacs_step.npz, iteration 0: dacs.py:274-315,357-860 on the CPU in fp32): losses, accuracies,
    pseudo-labels, confident-pixel count, the mixed tensors and the accumulated gradients -- reported and bounded at about twice the
    measured distance (bf16 met only the oracle before).  Behind the first optimizer step a bf16 run and the fp32
    reference are different trajectories, so only iteration 0 is compared."""
    from conftest import Target
    from cmda_amd import _lib
    from cmda_amd.optim import FlatAdamW
    from weights import DACS_CH, DACS_DIMS, DACS_SEEDS, DACS_SEG_SCALE, dacs_batch, sample_grad
    _lib._unbind_for_tests()
    if not torch.cuda.is_available():
        pytest.skip('no GPU on this machine')
    tgt = Target('gpu')
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(HERE, 'golden', 'dacs_step.npz')).items()}
    rt.set_compute_dtype(torch.bfloat16)
    try:
        dacs = build_train_model(make_cfg(DACS_DIMS, DACS_CH))
        seeded_fill(dacs.model, DACS_SEEDS['student'])
        seeded_fill(dacs.ema_model, DACS_SEEDS['teacher'])
        seeded_fill(dacs.cyclegan_itrd2en, DACS_SEEDS['generator'])
        with torch.no_grad():
            dacs.model.decode_head.conv_seg.weight.mul_(DACS_SEG_SCALE)
        dacs.to(tgt.device).train()
        opt = FlatAdamW(dacs.model, lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01)
        src, tg = dacs_batch()
        kmax = dacs._kmax()
        it = 0
        cj, bl, sigma = [float(v) for v in g[f'it{it}.gates']]
        cls = torch.full((1, kmax), -1, dtype=torch.int64)
        cls[0, :g[f'it{it}.classes'].numel()] = g[f'it{it}.classes']
        dacs.inject_draws = dict(choice=float(g[f'it{it}.choice']), color_jitter=cj, blur=bl, sigma=sigma, classes=cls, jitter=None,
                                 direction=[['leftdown', 'leftup'], ['rightdown', 'rightup']][int(cj * 10) % 2][int(cj * 100) % 2])
        batch = dict(source={k: tgt.to(v.clone()) for k, v in src.items()}, target={k: tgt.to(v.clone()) for k, v in tg.items()})
        res = dacs.train_step(batch, opt)
        torch.cuda.synchronize()
        lv, mix = res['log_vars'], dacs.last_mix
        ref_l = g[f'it{it}.losses'].float()
        got = torch.tensor([float(lv['decode.loss_seg']), float(lv['mix.decode.loss_seg'])])
        loss_err = ((got - ref_l[[0, 2]]).abs() / ref_l[[0, 2]].abs()).max().item()
        accs = torch.tensor([float(lv['decode.acc_seg']), float(lv['mix.decode.acc_seg'])])
        acc_err = (accs - ref_l[[1, 3]]).abs().max().item()
        agree = (mix['pseudo_label'].cpu().to(torch.uint8) == g[f'it{it}.pseudo_label']).float().mean().item()
        conf = abs(int(mix['pseudo_count']) - int(g[f'it{it}.pseudo_conf']))
        same = (mix['mixed_lbl'].cpu().to(torch.uint8) == g[f'it{it}.mixed_lbl']).float().mean().item()

        def rel(a, b):
            return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-30)).item()
        e_img = rel(mix['mixed_img'].cpu()[..., ::4, ::4], g[f'it{it}.mixed_img_s'])
        e_evt = rel(mix['mixed_events'].cpu()[:, :1, ::4, ::4], g[f'it{it}.mixed_events_s'])
        e_isr = rel(mix['mixed_isr'].cpu()[:, :1, ::2, ::2], g[f'it{it}.mixed_isr_s'])
        errs = []
        for k, p in dacs.model.named_parameters():
            ref_f = g[f'it{it}.grad.{k}']
            got_f = sample_grad(p.grad.cpu(), 24)
            errs.append(max((got_f[:-2] - ref_f[:-2]).abs().max().item() / (ref_f[:-2].abs().max().item() + 1e-12),
                            (got_f[-2:] - ref_f[-2:]).abs().max().item() / (ref_f[-1].abs().item() + 1e-12)))
        errs.sort()
        print(f'[bf16 vs reference step, iteration 0] losses {got.tolist()} vs {ref_l[[0, 2]].tolist()} (rel {loss_err:.2e}); accuracies abs err '
              f'{acc_err:.3f}; pseudo-labels {agree:.5f}; confident-pixel count difference {conf}; mixed-label agreement {same:.5f}; mixed '
              f'image {e_img:.2e}, events {e_evt:.2e}, ISR {e_isr:.2e}; gradient fingerprints median {errs[len(errs) // 2]:.2e}, 90th pct '
              f'{errs[int(len(errs) * 0.9)]:.2e}, worst {errs[-1]:.2e}')
        check_le('bf16 losses vs reference step (rel)', loss_err, BF16_FIXTURE['loss'])
        check_le('bf16 accuracies vs reference step (abs, percent)', acc_err, BF16_FIXTURE['acc'])
        check_ge('bf16 pseudo-label agreement with the reference step', agree, BF16_FIXTURE['labels'])
        check_le('bf16 confident-pixel count difference', conf, BF16_FIXTURE['conf'])
        check_ge('bf16 mixed-label agreement with the reference step', same, BF16_FIXTURE['mixed_labels'])
        check_le('bf16 mixed image vs reference step', e_img, 1e-5)            # (fp32 arithmetic in every mode)
        check_le('bf16 mixed events vs reference step', e_evt, BF16_FIXTURE['events'])
        check_le('bf16 mixed ISR vs reference step', e_isr, 1e-3)              # (uint8 luma of the fp32 mixed image)
        check_le('bf16 90th-percentile gradient fingerprint error vs reference step', errs[int(len(errs) * 0.9)], BF16_FIXTURE['grad_p90'])
        check_le('bf16 worst gradient fingerprint error vs reference step', errs[-1], BF16_FIXTURE['grad_worst'])
    finally:
        dacs.inject_draws = None
        rt.set_compute_dtype(torch.float32)


@pytest.mark.gpu
def test_dacs_overlapped_optimizer_update_matches_in_order_update_gpu():
    """FlatAdamW.overlap: AdamW, the gradient clear and the EMA update of the step boundary (dacs.py:250-315) on the optimizer's own
    stream, the weight-free head of the next captured iteration (mixing, frozen generator) underneath them, `runtime.wait_external` in
    front of the first use of a trainable weight.  Five train steps (one eager, four replayed) with the update in stream order and
    overlapped must give the same trajectory: same draws, losses and pseudo-labels; parameters / EMA weights / Adam moments equal up to
    the order of the fp32 atomics in the weight gradients."""
    from conftest import Target
    from cmda_amd import _lib
    from cmda_amd.optim import FlatAdamW
    _lib._unbind_for_tests()
    if not torch.cuda.is_available():
        pytest.skip('no GPU on this machine')
    tgt = Target('gpu')
    rt.set_compute_dtype(torch.float32)
    try:
        res = {}
        for overlap in (False, 'again', True):   # 'again': a second in-order run -- the run-to-run spread of the fp32 atomics, amplified by AdamW
            dacs = build_train_model(make_cfg(SMALL['dims'], SMALL['ch']))
            seeded_fill(dacs.model, 7)
            seeded_fill(dacs.ema_model, 8)
            seeded_fill(dacs.cyclegan_itrd2en, 9)
            dacs.to(tgt.device).train()
            opt = FlatAdamW(dacs.model, lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01)
            opt.overlap = overlap is True
            dacs.attach_flat_store(opt)
            src, tg = make_batch(2, 64, 64)
            batch = dict(source={k: tgt.to(v) for k, v in src.items()}, target={k: tgt.to(v) for k, v in tg.items()})
            torch.manual_seed(11), random.seed(11), np.random.seed(11)
            dacs.enable_graph(warmup_iters=1)
            losses, labels = [], []
            for it in range(5):
                out = dacs.train_step(batch, opt)
                losses.append([float(out['log_vars']['decode.loss_seg']), float(out['log_vars']['mix.decode.loss_seg'])])
                labels.append(dacs.last_mix['pseudo_label'].clone())
            opt.synchronize()
            torch.cuda.synchronize()
            assert dacs._graph is not None
            res[overlap] = dict(losses=torch.tensor(losses), labels=[t.cpu() for t in labels], p=opt.flat_p.cpu().clone(),
                                m=opt.flat_m.cpu().clone(), ema=dacs._flat[1].cpu().clone(), g=opt.flat_g.cpu().clone())
        a, a2, b = res[False], res['again'], res[True]
        assert_close(b['losses'], a['losses'], 2e-4, name='losses, overlapped update vs in-order update')
        for it, (x, y) in enumerate(zip(a['labels'], b['labels'])):
            check_ge(f'it{it} pseudo-label agreement, overlapped vs in-order update', (x == y).float().mean().item(), 0.999)
        assert_close(b['p'], a['p'], 1e-5, atol=5 * 6e-5 * 1.5, name='parameters after five steps')   # (AdamW: +-lr per step on round-off-level gradients)
        assert_close(b['ema'], a['ema'], 1e-5, atol=5 * 6e-5 * 1.5, name='EMA teacher after five steps')
        # the last step's gradients: two in-order runs already differ -- atomic order -> +-lr steps of AdamW on round-off-level gradients ->
        # four steps of drift, the effect documented for the reference fixture above -- and by how much varies from run to run (observed
        # over six fresh-box runs: in-order vs in-order 1.5e-3 .. 1.9e-2, overlapped vs in-order 1.2e-3 .. 2.7e-2 of the largest
        # gradient): both are printed, the overlapped run is bounded by a loose absolute figure.  What would betray a mis-ordered update
        # is the loss / parameter comparison above (a stale weight moves the losses by 1e-2), not this number.
        scale = a['g'].abs().max().item()
        spread = (a2['g'] - a['g']).abs().max().item() / scale
        got = (b['g'] - a['g']).abs().max().item() / scale
        print(f'last-step gradients: overlapped vs in-order {got:.2e}, in-order vs in-order {spread:.2e} (of the largest gradient)')
        check_le('last-step gradients, overlapped vs in-order update (of the largest gradient)', got, 0.1)
    finally:
        rt.set_compute_dtype(torch.float32)


class _MaskFeed(torch.nn.Module):
    """stands in for an oracle DropPath: pops the next per-sample keep factors (already divided by keep)"""

    def __init__(self, queue):
        super().__init__()
        self.queue = queue

    def forward(self, x):
        m = self.queue.pop(0)
        return x * m.view(-1, *([1] * (x.dim() - 1))).to(x.dtype)


class _Dropout2dFeed(torch.nn.Module):
    def __init__(self, queue, keep):
        super().__init__()
        self.queue, self.keep = queue, keep

    def forward(self, x):
        return x * (self.queue.pop(0) / self.keep)[:, :, None, None]


def _feed_oracle_masks(ref, taps, B, keep):
    """Route the masks one HIP pass drew (taps: rows 2i / 2i+1 = attention / MLP branch of live block i, columns = the samples of
    the joint pass) to the oracle's modules in the order its two forward_train calls consume them.  Image encoder / fusion
    blocks / Dropout2d: columns [source | mixed]; event encoder (one 4B batch on the GPU): [source events | mixed events | source
    ISR | mixed ISR], while the oracle runs events then ISR inside the source step, then again inside the mixed step."""
    def live(blocks):
        return [b for b in blocks if not isinstance(b.drop_path, torch.nn.Identity)]

    def wire(blocks, masks, col_sets):
        for i, blk in enumerate(live(blocks)):
            q = []
            for cols in col_sets:
                q += [masks[2 * i][cols].cpu(), masks[2 * i + 1][cols].cpu()]
            blk.drop_path = _MaskFeed(q)
    s = lambda k: slice(k * B, (k + 1) * B)  # noqa: E731
    enc = lambda m: [b for st in range(1, 5) for b in getattr(m, f'block{st}')]  # noqa: E731
    wire(enc(ref.backbone_image), taps[('drop_path', 'image')], [s(0), s(1)])
    wire(enc(ref.backbone_events), taps[('drop_path', 'events')], [s(0), s(2), s(1), s(3)])
    wire(list(ref.fusion_module.basic_block), taps[('drop_path', 'fusion')], [s(0), s(1)])
    d = taps[('dropout2d', 'head')].cpu()
    ref.decode_head.dropout = _Dropout2dFeed([d[s(0)], d[s(1)]], keep)


@pytest.mark.gpu
def test_dacs_graph_replay_draws_fresh_masks_and_matches_oracle():
    """VERDICT r02 #4b: with DropPath / Dropout2d ON (rates well above the recipe's so that every replay drops something), the
    segmented hipGraph replay must draw FRESH masks per replay (torch's philox bookkeeping across 13 separately captured
    segments) and the replayed iteration must equal the oracle's iteration fed the very masks that replay drew (read back from
    the tensors the kernels consumed, runtime.taps)."""
    from conftest import Target
    from cmda_amd import _lib
    _lib._unbind_for_tests()
    if not torch.cuda.is_available():
        pytest.skip('no GPU on this machine')
    tgt = Target('gpu')
    rt.set_compute_dtype(torch.float32)
    dims, ch, B, S = SMALL['dims'], SMALL['ch'], 2, 64
    dpr, fdp, drop = 0.4, 0.3, 0.3
    kw = dict(drop_path_rate=dpr, fusion_drop_path=fdp, dropout_ratio=drop)
    dacs = build_train_model(make_cfg(dims, ch, **kw))
    seeded_fill(dacs.model, 7)
    seeded_fill(dacs.ema_model, 8)
    seeded_fill(dacs.cyclegan_itrd2en, 9)
    dacs.to(tgt.device).train()
    st = dacs.model
    st.backbone_image._tap_name, st.backbone_events._tap_name, st.fusion_module._tap_name, st.decode_head._tap_name = 'image', 'events', 'fusion', 'head'
    src, tg = make_batch(B, S, S)
    batch = dict(source={k: tgt.to(v) for k, v in src.items()}, target={k: tgt.to(v) for k, v in tg.items()})
    ref, ema, G = oracle_student(dims, ch, **kw), oracle_student(dims, ch), ocg.ResnetGenerator().eval()
    seeded_fill(ref, 7).train()
    seeded_fill(ema, 8).train()
    seeded_fill(G, 9)
    torch.manual_seed(11), random.seed(11), np.random.seed(11)
    dacs.enable_graph(warmup_iters=1)
    rt.taps = {}
    try:
        seen = []
        for it in range(4):
            for p in dacs.model.parameters():
                if p.grad is not None:
                    p.grad.zero_()
            log_vars = dacs(**batch)
            torch.cuda.synchronize()
            assert (dacs._graph is not None) == (it >= 1)
            # the student tensors only (the teacher passes run with the stochastic layers off and publish nothing)
            taps = {k: v.detach().clone() for k, v in rt.taps.items()}
            assert set(taps) == {('drop_path', 'image'), ('drop_path', 'events'), ('drop_path', 'fusion'), ('dropout2d', 'head')}
            assert taps[('drop_path', 'image')].shape[1] == 2 * B and taps[('drop_path', 'events')].shape[1] == 4 * B
            seen.append(taps)
            grads = {n: p.grad.detach().cpu().clone() for n, p in dacs.model.named_parameters()}
            for p in ref.parameters():
                p.grad = None
            _feed_oracle_masks(ref, taps, B, 1 - drop)
            o = dacs_iter.dacs_iteration(ref, ema, G, src, tg, local_iter=it, forward_cfg=FCFG, isr_parms=ISR, shift_type='random',
                                         draws=oracle_draws(dacs.last_draws))
            for blk in [b for m in (ref.backbone_image, ref.backbone_events) for s_ in range(1, 5) for b in getattr(m, f'block{s_}')]:
                assert isinstance(blk.drop_path, torch.nn.Identity) or not blk.drop_path.queue, 'oracle left injected masks unused'
            out = ({k: v.detach().clone() for k, v in log_vars.items()}, dacs.last_mix, grads, o, {n: q.grad.clone() for n, q in ref.named_parameters()})
            check_iteration(out, True, 1e-4, 0.15)   # (gradients of a pass with 40 % DropPath: few samples carry each branch; 5.1e-2 measured)
            print(f'iteration {it} ({"replay" if it >= 1 and dacs._graph is not None else "eager"}): source loss '
                  f'{log_vars["decode.loss_seg"].item():.6f} vs {o["decode.loss_seg"].item():.6f}; dropped entries '
                  f'{int((taps[("drop_path", "events")] == 0).sum())} / {taps[("drop_path", "events")].numel()}')
        # replays 2 and 3 re-run the captured launches of iteration 1: their masks must differ from it and from each other
        for k in seen[1]:
            assert not torch.equal(seen[1][k], seen[2][k]) and not torch.equal(seen[2][k], seen[3][k]), f'{k}: the replay repeated its masks'
            assert (seen[2][k] == 0).any() and (seen[2][k] > 0).any()
    finally:
        rt.taps = None


# ---------------------------------------------------------------------------------------------------------------------------
# The HIP step against the reference's OWN DACS.train_step (tests/golden/dacs_step.npz, written by make_golden.py::dacs_step from
# mmseg/models/uda/dacs.py:274-315,357-860 run on the CPU for local_iter 0, 1, 2 -- SURVEY 8c items 7 and 11).  The oracle is not
# involved: the fixture holds what the reference produced.
@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['f32', 'x3'])
def test_dacs_train_step_against_reference_fixture_gpu(mode):
    """three train_step calls (EMA init / update, source step, teacher + pseudo-labels, ClassMix + ISR of the mixed image, mixed
    step, FlatAdamW) on one 512 x 512 pair with the reference's recorded draws injected; compared per iteration: losses, pseudo-labels,
    confident-pixel count, mixed image / events / ISR / label / pseudo-weight, accumulated gradients, EMA teacher, student after the
    optimizer step, BatchNorm running statistics; then _update_ema(1500)."""
    from conftest import Target
    from cmda_amd import _lib
    from cmda_amd.optim import FlatAdamW
    from weights import DACS_CH, DACS_DIMS, DACS_SEEDS, DACS_SEG_SCALE, dacs_batch, sample_grad
    _lib._unbind_for_tests()
    if not torch.cuda.is_available():
        pytest.skip('no GPU on this machine')
    tgt = Target('gpu')
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(HERE, 'golden', 'dacs_step.npz')).items()}
    rt.set_compute_dtype(torch.float32)
    rt.set_gemm_x3(mode == 'x3')
    try:
        dacs = build_train_model(make_cfg(DACS_DIMS, DACS_CH))
        seeded_fill(dacs.model, DACS_SEEDS['student'])
        seeded_fill(dacs.ema_model, DACS_SEEDS['teacher'])
        seeded_fill(dacs.cyclegan_itrd2en, DACS_SEEDS['generator'])
        with torch.no_grad():
            dacs.model.decode_head.conv_seg.weight.mul_(DACS_SEG_SCALE)
        dacs.to(tgt.device).train()
        opt = FlatAdamW(dacs.model, lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01)
        src, tg = dacs_batch()
        kmax = dacs._kmax()
        exact = mode == 'f32'
        for it in range(3):
            cj, bl, sigma = [float(v) for v in g[f'it{it}.gates']]
            cls = torch.full((1, kmax), -1, dtype=torch.int64)
            cls[0, :g[f'it{it}.classes'].numel()] = g[f'it{it}.classes']
            dacs.inject_draws = dict(choice=float(g[f'it{it}.choice']), color_jitter=cj, blur=bl, sigma=sigma, classes=cls, jitter=None,
                                     direction=[['leftdown', 'leftup'], ['rightdown', 'rightup']][int(cj * 10) % 2][int(cj * 100) % 2])
            batch = dict(source={k: tgt.to(v.clone()) for k, v in src.items()}, target={k: tgt.to(v.clone()) for k, v in tg.items()})
            res = dacs.train_step(batch, opt)
            torch.cuda.synchronize()
            lv, mix = res['log_vars'], dacs.last_mix
            ref_l = g[f'it{it}.losses'].float()
            got = torch.tensor([float(lv['decode.loss_seg']), float(lv['mix.decode.loss_seg'])])
            tol_l = (1e-4 if exact else 3e-4) * (1 if it == 0 else 10)
            assert_close(got, ref_l[[0, 2]], tol_l, name=f'it{it} losses vs reference')
            accs = torch.tensor([float(lv['decode.acc_seg']), float(lv['mix.decode.acc_seg'])])
            assert_close(accs, ref_l[[1, 3]], 2e-3, name=f'it{it} accuracies vs reference')
            agree = (mix['pseudo_label'].cpu().to(torch.uint8) == g[f'it{it}.pseudo_label']).float().mean().item()
            check_ge(f'it{it} pseudo-label agreement with the reference', agree, 0.999)
            check_le(f'it{it} confident-pixel count difference', abs(int(mix['pseudo_count']) - int(g[f'it{it}.pseudo_conf'])), 150)
            assert_close(mix['mixed_img'].cpu()[..., ::4, ::4], g[f'it{it}.mixed_img_s'], 1e-6, name=f'it{it} mixed image')
            assert_close(mix['mixed_events'].cpu()[:, :1, ::4, ::4], g[f'it{it}.mixed_events_s'], 5e-4, atol=1e-4, name=f'it{it} mixed events')
            assert_close(mix['mixed_isr'].cpu()[:, :1, ::2, ::2], g[f'it{it}.mixed_isr_s'].float(), 1e-3, name=f'it{it} mixed ISR')
            same = (mix['mixed_lbl'].cpu().to(torch.uint8) == g[f'it{it}.mixed_lbl']).float().mean().item()
            check_ge(f'it{it} mixed-label agreement with the reference', same, 0.999)
            assert_close(mix['pseudo_weight'].cpu()[..., ::8, ::8], g[f'it{it}.mixed_weight_s'], 2e-3, name=f'it{it} mixed pseudo-weight')
            # gradients (the fixture's fingerprints: strided sample + sum + abs-sum per tensor); AdamW has already consumed them but
            # leaves them in place.  Bound on the worst tensor, relative to the tensor's own scale.
            worst, seen = 0.0, 0
            for k, p in dacs.model.named_parameters():
                ref_f = g[f'it{it}.grad.{k}']
                got_f = sample_grad(p.grad.cpu(), 24)   # sample relative to its largest element, sums relative to the abs-sum
                d = max((got_f[:-2] - ref_f[:-2]).abs().max().item() / (ref_f[:-2].abs().max().item() + 1e-12),
                        (got_f[-2:] - ref_f[-2:]).abs().max().item() / (ref_f[-1].abs().item() + 1e-12))
                worst, seen = max(worst, d), seen + 1
            assert seen == sum(k.startswith(f'it{it}.grad.') for k in g)
            # iteration 0: fp32 round-off (3.8e-3 measured; split-bf16 3.2e-3 .. 1.2e-2); behind the first optimizer step the runs differ by
            # AdamW's +-lr noise on the zero-gradient parameters and a handful of flipped pseudo-labels.  Exact fp32: 2.8e-2 .. 4.2e-2 at
            # iteration 1, up to 7.5e-2 at iteration 2 (seven runs); split-bf16: 4e-2 .. 7.6e-2 at iteration 1, 6.6e-2 .. 0.157 at iteration 2
            # (14 runs of four kernel configurations, tools/gpu/fixture_var.sh) -- the spread is the same with and without the fused
            # BatchNorm statistics and the three-launch gate: it is the order of the fp32 atomics amplified by two optimizer steps
            check_le(f'it{it} worst gradient fingerprint error vs reference', worst,
                     (1.2e-2 if exact else 3e-2) if it == 0 else (0.3 if exact else 0.4))   # (worst of ten fp32 runs 0.108, of seventeen split-bf16 runs 0.157)
            for k, p in dacs.model.named_parameters():
                # (AdamW turns the round-off-level gradients of the key half of every kv.bias into +-lr steps of arbitrary sign: up to
                # 32 elements x 6e-5 per iteration on the fingerprint's sums)
                assert_close(sample_grad(p.data.cpu(), 24), g[f'it{it}.param.{k}'], 2e-5, atol=8e-4 * (it + 1), name=f'it{it} param {k}')
            for k, p in dacs.ema_model.named_parameters():
                assert_close(sample_grad(p.data.cpu(), 24), g[f'it{it}.ema.{k}'], 2e-5, atol=8e-4 * it if it else 1e-6, name=f'it{it} ema {k}')
            for k, b in dacs.model.named_buffers():
                if f'it{it}.bn.{k}' in g:
                    assert_close(b.cpu(), g[f'it{it}.bn.{k}'], 5e-4 if it == 0 else 8e-3, atol=1e-5, name=f'it{it} {k}')   # (behind the optimizer steps: 3.2e-3 worst of two runs)
            print(f'iteration {it}: losses {got.tolist()} vs reference {ref_l[[0, 2]].tolist()}, pseudo-labels {agree:.5f}, worst gradient {worst:.2e}')
        assert dacs.local_iter == 3
        dacs._update_ema(1500)
        torch.cuda.synchronize()
        for k, p in dacs.ema_model.named_parameters():
            assert_close(sample_grad(p.data.cpu(), 24), g[f'ema1500.{k}'], 2e-5, atol=2.4e-3, name=f'ema1500 {k}')
    finally:
        dacs.inject_draws = None
        rt.set_gemm_x3(False)
        rt.set_compute_dtype(torch.float32)
