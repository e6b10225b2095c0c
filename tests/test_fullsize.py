"""Parity where the bench runs: full-depth MiT-B5 at 512x512 (stage grids 128/64/32/16, Nk = 256 in every stage: the fused
attention kernel and the large GEMM tiles in bf16 mode), the full image+events fusion student, the stochastic paths (DropPath,
Dropout2d) with injected masks, and the pseudo-label kernel bit for bit on the GPU.

The oracle (oracle/, CPU fp32) costs seconds per 512x512 image on the GPU box's host, so these tests are `gpu`-marked except
the injected-mask test, which is small and also runs in the emulator.
  fp32 mode: north-star bound 1e-3 relative (max-norm) on logits.
  bf16 mode: the error of the 52-block model is REPORTED and bounded (logits within 6e-2 of the logit range; argmax
  agreement >= 97 %), which is what the throughput numbers are quoted at.
"""
import functools
import os
import sys

import pytest
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
from weights import seeded_fill, seeded_randn  # noqa: E402

import cmda_amd  # noqa: E402,F401
import cmda_amd.runtime as rt  # noqa: E402
from cmda_amd import ops  # noqa: E402
from cmda_amd.registry import build_segmentor  # noqa: E402
from conftest import Target, assert_close, check_ge, check_le  # noqa: E402
from oracle import fusion as ofu, head as ohd, mit as omit, segmentor as oseg, uda as ouda  # noqa: E402

DIMS = [64, 128, 320, 512]
DECODER = dict(embed_dims=256, embed_cfg=dict(type='mlp', act_cfg=None, norm_cfg=None),
               embed_neck_cfg=dict(type='mlp', act_cfg=None, norm_cfg=None),
               fusion_cfg=dict(type='aspp', sep=True, dilations=(1, 6, 12, 18), pool=False, act_cfg=dict(type='ReLU'),
                               norm_cfg=dict(type='BN', requires_grad=True)))
HEAD = dict(in_channels=DIMS, in_index=[0, 1, 2, 3], channels=256, num_classes=19, norm_cfg=dict(type='BN', requires_grad=True),
            align_corners=False, loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0))
FCFG = dict(loss_weight={'image': 0.5, 'events': 0.5, 'fusion': 0.5, 'img_self_res': 0.25}, gradual_rate=0.0)


def gpu_target():
    from cmda_amd import _lib
    _lib._unbind_for_tests()
    if not torch.cuda.is_available():
        pytest.skip('no GPU on this machine')
    return Target('gpu')


def labels(B, S, seed):
    g = torch.Generator().manual_seed(seed)
    lab = torch.randint(0, 19, (B, 1, S // 32, S // 32), generator=g).repeat_interleave(32, 2).repeat_interleave(32, 3)
    lab[torch.rand(B, 1, S, S, generator=g) < 0.05] = 255
    return lab


def rel(a, b):
    return (a.detach().float().cpu() - b.detach().float()).abs().max().item() / (b.detach().abs().max().item() + 1e-30)


def grad_errors(model, ref):
    """max-norm relative error of every parameter gradient, sorted"""
    errs = []
    for (n1, p), (n2, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert n1 == n2
        if q.grad is not None:
            errs.append(rel(p.grad, q.grad))
    return sorted(errs)


class ReluFlipProbe:
    """Makes the 'a train-mode BatchNorm + ReLU pre-activation within round-off of zero may flip its mask' caveat CHECKABLE
    (VERDICT r02 #4d): hooks every BatchNorm2d of the oracle's decode head (each feeds a ReLU) and records, per layer, the channels
    that hold a pre-activation within `tau` of zero -- the only places where a different fp32 summation order of the batch
    statistics can change the ReLU mask.  `masked_errors` then compares every parameter gradient with those channels of THAT
    layer's BatchNorm affine and of the convolution feeding it left out; everything else must agree tightly."""

    def __init__(self, head, tau=4e-6):
        self.tau, self.suspect, self.total = tau, {}, 0
        for name, m in head.named_modules():
            if isinstance(m, nn.BatchNorm2d):
                m.register_forward_hook(self._hook(name))

    def _hook(self, name):
        def fn(mod, inp, out):
            y = out.detach()
            near = (y.abs() < self.tau * max(1.0, y.abs().max().item() / 4.0))
            ch = near.any(dim=0).any(dim=-1).any(dim=-1).nonzero().flatten().tolist()
            self.suspect.setdefault(name, set()).update(ch)
            self.total += int(near.sum())
        return fn

    def mask_for(self, pname, tensor, prefix='decode_head.'):
        """boolean keep-mask over dim 0 of a parameter gradient (None = keep everything)"""
        if not pname.startswith(prefix):
            return None
        rel_name = pname[len(prefix):]
        for bn_name, chans in self.suspect.items():
            owner = bn_name[:-len('.bn')] if bn_name.endswith('.bn') else None
            if owner is None or not chans:
                continue
            if rel_name in (owner + '.bn.weight', owner + '.bn.bias', owner + '.conv.weight'):
                keep = torch.ones(tensor.shape[0], dtype=torch.bool)
                keep[sorted(chans)] = False
                return keep
        return None

    def masked_errors(self, model, ref, prefix='decode_head.'):
        out = []
        for (n1, p), (n2, q) in zip(model.named_parameters(), ref.named_parameters()):
            assert n1 == n2
            if q.grad is None:
                continue
            g, r = p.grad.detach().cpu().float(), q.grad.float()
            raw = ((g - r).abs().max() / (r.abs().max() + 1e-12)).item()
            keep = self.mask_for(n1, r, prefix)
            if keep is not None and keep.any():
                m = ((g[keep] - r[keep]).abs().max() / (r[keep].abs().max() + 1e-12)).item()
            else:
                m = raw
            out.append((m, raw, n1, 0 if keep is None else int((~keep).sum())))
        return sorted(out, reverse=True)


def grad_report(model, ref):
    """worst max-norm relative error over the parameter gradients + the fraction of tensors within 2e-2"""
    errs = grad_errors(model, ref)
    return errs[-1], sum(e < 2e-2 for e in errs) / max(len(errs), 1)


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_mit_b5_daformer_512_vs_oracle(mode):
    """BASELINE configs[1] at its real size: full MiT-B5 (52 blocks) + DAFormerHead, one 512x512 image, fwd + bwd."""
    tgt = gpu_target()
    rt.set_compute_dtype(torch.float32 if mode == 'f32' else torch.bfloat16)
    cfg = dict(type='EncoderDecoder', backbone=dict(type='mit_b5', style='pytorch', drop_path_rate=0.0),
               decode_head=dict(type='DAFormerHead', dropout_ratio=0.0, decoder_params=dict(DECODER), **HEAD))
    model = build_segmentor(cfg)
    torch.manual_seed(5)
    model.init_weights()
    ref = oseg.EncoderDecoder(omit.mit_b5(drop_path_rate=0.0), ohd.DAFormerHead(dropout_ratio=0.0))
    ref.load_state_dict(model.state_dict())
    model.to(tgt.device).train()
    ref.train()
    probe = ReluFlipProbe(ref.decode_head)
    img, gt = seeded_randn((1, 3, 512, 512), 5, 'img'), labels(1, 512, 5)
    losses, logits = model.forward_train(tgt.to(img), None, tgt.to(gt))
    losses['decode.loss_seg'].backward()
    rl, rlog = ref.forward_train(img, gt)
    rl['decode.loss_seg'].backward()
    torch.cuda.synchronize()
    e_log = rel(logits, rlog)
    agree = (logits.argmax(1).cpu() == rlog.argmax(1)).float().mean().item()
    worst, within = grad_report(model, ref)
    print(f'[{mode}] 512x512 MiT-B5+DAFormerHead: logits rel err {e_log:.3e}, argmax agreement {agree:.4f}, loss '
          f'{losses["decode.loss_seg"].item():.6f} vs {rl["decode.loss_seg"].item():.6f}, worst grad rel err {worst:.3e}, '
          f'{within:.1%} of gradient tensors within 2e-2')
    if mode == 'f32':
        check_le('fp32 logits rel err', e_log, 1e-3, strict=True)          # the north star's tolerance
        check_le('fp32 logits rel err [regression gate]', e_log, 1e-5)   # 2.6e-6 measured (profiles/r04_test_margins.txt)
        assert_close(losses['decode.loss_seg'], rl['decode.loss_seg'], 1e-4, name='loss')
        # gradients: every tensor within 2e-2 once the channels whose BatchNorm + ReLU pre-activation sits within round-off of zero
        # (a flipped mask bit moves that channel's gradient) are left out of THEIR layer's comparison -- no blanket 0.2 bound
        me = probe.masked_errors(model, ref)
        nsus = sum(len(v) for v in probe.suspect.values())
        print(f'[{mode}] {probe.total} pre-activations within round-off of zero in {nsus} (layer, channel) pairs; worst masked gradient errors: '
              + '; '.join(f'{n} {m:.2e} (raw {r:.2e}, {k} channels out)' for m, r, n, k in me[:6]))
        check_le('worst masked gradient rel err', me[0][0], 2e-2, strict=True)
        assert nsus < 0.25 * sum(m.num_features for m in ref.decode_head.modules() if isinstance(m, nn.BatchNorm2d)), 'probe masks too much'
    else:
        check_le('bf16 logits rel err', e_log, 3e-2, strict=True)          # (1.0e-2 measured, three boxes)
        check_ge('bf16 argmax agreement', agree, 0.96, strict=True)   # (0.9875-0.9884 measured)
        assert_close(losses['decode.loss_seg'], rl['decode.loss_seg'], 2e-3, name='loss')   # (1e-5 measured)
    rt.set_compute_dtype(torch.float32)


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_fusion_student_512_vs_oracle(mode):
    """The bench's student at full width / depth / size: two MiT-B5 encoders (events + ISR batched through the event encoder),
    AttentionAvgFusion, the shared DAFormerHeadFusion run jointly over the four feature sets, four CE terms, fwd + bwd."""
    tgt = gpu_target()
    rt.set_compute_dtype(torch.float32 if mode == 'f32' else torch.bfloat16)
    bbc = dict(type='mit_b5', style='pytorch', drop_path_rate=0.0)
    head = dict(type='DAFormerHeadFusion', dropout_ratio=0.0,
                decoder_params=dict(DECODER, train_type='cs2dsec_image+events_together', share_decoder=True), **HEAD)
    model = build_segmentor(dict(type='FusionEncoderDecoder', backbone_image=dict(bbc), backbone_events=dict(bbc),
                                 fusion_module=dict(type='AttentionAvgFusion', in_channels=DIMS, drop_path_rate=0.0),
                                 decode_head=head, train_type='cs2dsec_image+events_together'))
    torch.manual_seed(6)
    model.init_weights()
    ref = oseg.FusionEncoderDecoder(backbone_image=omit.mit_b5(drop_path_rate=0.0), backbone_events=omit.mit_b5(drop_path_rate=0.0),
                                    fusion_module=ofu.AttentionAvgFusion(drop_path_rate=0.0),
                                    decode_head=ohd.DAFormerHeadFusion(dropout_ratio=0.0, share_decoder=True))
    ref.load_state_dict(model.state_dict())
    model.to(tgt.device).train()
    ref.train()
    S = 512
    inp = dict(image=seeded_randn((1, 3, S, S), 6, 'img'), events=seeded_randn((1, 3, S, S), 6, 'ev').clamp(-1, 1),
               img_self_res=seeded_randn((1, 3, S, S), 6, 'isr').clamp(-1, 1))
    gt = labels(1, S, 6)
    wgt = torch.rand(1, S, S, generator=torch.Generator().manual_seed(6))
    losses, pred = model.forward_train({k: tgt.to(v) for k, v in inp.items()}, tgt.to(gt), seg_weight=tgt.to(wgt), cfg=FCFG)
    losses['decode.loss_seg'].backward()
    rl, rpred = ref.forward_train(inp, gt, seg_weight=wgt, cfg=FCFG)
    rl['decode.loss_seg'].backward()
    torch.cuda.synchronize()
    errs = {k: rel(pred[k], rpred[k]) for k in pred}
    worst, within = grad_report(model, ref)
    print(f'[{mode}] 512x512 fusion student: logits rel err {errs}, loss {losses["decode.loss_seg"].item():.6f} vs '
          f'{rl["decode.loss_seg"].item():.6f}, worst grad rel err {worst:.3e}, {within:.1%} of gradient tensors within 2e-2')
    if mode == 'f32':
        check_le('fp32 logits rel err (worst branch)', max(errs.values()), 1e-3, strict=True)   # the north star's tolerance
        check_le('fp32 logits rel err (worst branch) [regression gate]', max(errs.values()), 1e-5)   # 2.9e-6 measured
        assert_close(losses['decode.loss_seg'], rl['decode.loss_seg'], 1e-4, name='loss')
        check_ge('fraction of gradient tensors within 2e-2', within, 0.97, strict=True)
        check_le('worst gradient rel err', worst, 1e-2, strict=True)    # (3.2e-3 measured, three boxes)
    else:
        check_le('bf16 logits rel err (worst branch)', max(errs.values()), 3.5e-2, strict=True)   # (1.4e-2 measured)
        assert_close(losses['decode.loss_seg'], rl['decode.loss_seg'], 2e-3, name='loss')
    rt.set_compute_dtype(torch.float32)


class _InjectedDropPath(nn.Module):
    """stands in for the oracle's DropPath: pops the next injected per-sample keep mask (already scaled by 1/keep)"""

    def __init__(self, pool):
        super().__init__()
        self.pool = pool

    def forward(self, x):
        m = self.pool.pop(0)
        return x * m.view(-1, *([1] * (x.dim() - 1)))


class _InjectedDropout2d(nn.Module):
    def __init__(self, mask, keep):
        super().__init__()
        self.mask, self.keep = mask, keep

    def forward(self, x):
        return x * (self.mask / self.keep)[:, :, None, None]


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_stochastic_paths_with_injected_masks(tgt, mode):
    """DropPath (timm, per sample per residual branch: mix_transformer.py:134,145-146) and Dropout2d (decode_head.py:565-566) are
    ON in every training step; here both sides get the SAME masks, so the row-scale GEMM epilogue, sample_scale and their
    backward are compared against the oracle instead of being switched off."""
    if mode == 'bf16' and tgt.kind == 'emu':
        pytest.skip('bf16 variant on the GPU only (emulated bf16 MFMA is slow)')
    dt = torch.float32 if mode == 'f32' else torch.bfloat16
    rt.set_compute_dtype(dt)
    depths, dpr, drop = [2, 2, 2, 2], 0.4, 0.3
    cfg = dict(type='EncoderDecoder',
               backbone=dict(type='MixVisionTransformer', embed_dims=DIMS, num_heads=[1, 2, 5, 8], qkv_bias=True, depths=depths,
                             sr_ratios=[8, 4, 2, 1], drop_path_rate=dpr, norm_layer=functools.partial(nn.LayerNorm, eps=1e-6)),
               decode_head=dict(type='DAFormerHead', dropout_ratio=drop, decoder_params=dict(DECODER), **HEAD))
    model = build_segmentor(cfg)
    seeded_fill(model, 13)
    ref = oseg.EncoderDecoder(omit.MixVisionTransformer(depths=depths, drop_path_rate=dpr), ohd.DAFormerHead(dropout_ratio=drop))
    ref.load_state_dict(model.state_dict())
    model.to(tgt.device).train()
    ref.train()
    B, S = 3, 64
    g = torch.Generator().manual_seed(13)
    rates = [v.item() for v in torch.linspace(0, dpr, sum(depths))]
    live = [r for r in rates if r > 0]
    # per live block two masks (attention branch, MLP branch), each [B] of {0, 1/keep}
    masks = torch.stack([torch.floor((1 - r) + torch.rand(B, generator=g)) / (1 - r) for r in live for _ in range(2)])
    assert (masks == 0).any() and (masks > 0).any()
    keep = 1 - drop
    dmask = (torch.rand(B, 256, generator=g) < keep).float()

    def inject(B_, device):
        blocks = [blk for s in range(1, 5) for blk in getattr(model.backbone, f'block{s}')]
        mk = masks.to(device)
        i = 0
        for blk in blocks:
            if blk.drop_path_rate > 0:
                blk._dp_pool = [mk, 2 * i]
                i += 1
    model.backbone._draw_drop_path = inject
    model.decode_head.inject_dropout_mask = dmask
    pool = [m for m in masks]
    for s in range(1, 5):
        for blk in getattr(ref.backbone, f'block{s}'):
            if not isinstance(blk.drop_path, nn.Identity):
                blk.drop_path = _InjectedDropPath(pool)
    ref.decode_head.dropout = _InjectedDropout2d(dmask, keep)
    img, gt = seeded_randn((B, 3, S, S), 13, 'img'), labels(B, S, 13)
    losses, logits = model.forward_train(tgt.to(img), None, tgt.to(gt))
    losses['decode.loss_seg'].backward()
    rl, rlog = ref.forward_train(img, gt)
    rl['decode.loss_seg'].backward()
    assert not pool, 'the oracle did not consume every injected mask'
    tol = 1e-5 if mode == 'f32' else 5e-2   # (fp32 2.8e-6, bf16 1.9e-2 of the logit range measured)
    assert_close(logits, rlog, tol, name='logits with injected DropPath / Dropout2d masks')
    assert_close(losses['decode.loss_seg'], rl['decode.loss_seg'], 1e-4 if mode == 'f32' else 2e-3, name='loss')
    errs = grad_errors(model, ref)
    med, p90, worst = errs[len(errs) // 2], errs[int(len(errs) * 0.9)], errs[-1]
    print(f'[{mode}] injected masks: gradient rel err median {med:.2e}, 90th percentile {p90:.2e}, worst {worst:.2e}')
    if mode == 'f32':
        # (GPU, 128 x 128: 3.0e-4 / 4.4e-3 measured on three boxes; the emulator's 32 x 32 case: 1.9e-3 / 9e-3 -- fewer pixels per gradient)
        check_le('gradient rel err 90th pct', p90, 1e-3 if tgt.kind == 'gpu' else 5e-3, strict=True)
        check_le('gradient rel err worst', worst, 1.5e-2 if tgt.kind == 'gpu' else 3e-2, strict=True)
    else:   # bf16 activations: bounded in the bulk (tensors with tiny gradients carry large relative max-norm errors)
        check_le('bf16 gradient rel err median', med, 0.165, strict=True)   # (6.6e-2 worst of three boxes: 2.5x)
        check_le('bf16 gradient rel err 90th pct', p90, 0.26, strict=True)   # (0.101)
    rt.set_compute_dtype(torch.float32)


@pytest.mark.gpu
def test_pseudo_label_bit_exact_gpu():
    """north star: pseudo-label argmax bit-exact.  The kernel's up-sampling is the individually rounded bilinear formula of
    oracle.uda.upsample_exact; labels must be EQUAL (first-max tie rule included) and the confident-pixel count equal -- at the
    teacher's real size (2 x 19 x 128 x 128 -> 512 x 512) and at an odd ratio."""
    tgt = gpu_target()
    for seed, (B, h, w, H, W) in enumerate([(2, 128, 128, 512, 512), (1, 110, 160, 440, 640), (2, 13, 9, 37, 50)]):
        g = torch.Generator().manual_seed(100 + seed)
        logits = torch.randn(B, h, w, 19, generator=g) * 3
        logits[0, : h // 2, : w // 2, 4] = logits[0, : h // 2, : w // 2, 7]   # exact ties between two classes: first max wins
        up = ouda.upsample_exact(logits.permute(0, 3, 1, 2).contiguous(), (H, W))
        prob_ref, lab_ref = torch.softmax(up, 1).max(1)
        lab, prob, cnt = ops.pseudo_label(tgt.to(logits), H, W, 0.968)
        assert torch.equal(lab.cpu(), lab_ref), f'{(lab.cpu() != lab_ref).sum().item()} labels differ at {(B, h, w, H, W)}'
        assert cnt.item() == int((prob_ref >= 0.968).sum()), (cnt.item(), int((prob_ref >= 0.968).sum()))
        assert_close(prob, prob_ref, 1e-6, name='pseudo prob')


@pytest.mark.gpu
def test_fusion_simple_test_440x640_vs_oracle():
    """FusionEncoderDecoder.simple_test(warp_image, events_vg) at the DSEC evaluation size 440x640, FULL depth
    (encoder_decoder.py:897-984): token grids 110x160 / 55x80 / 28x40 / 14x20, Nk = 260 / 260 / 280 / 280 keys (beyond the fused
    attention kernel: the GEMM + soft-max path), the joint decoder pass with three feature sets, logits resized to the input size,
    soft-max, argmax; then mIoU of both label maps against the same ground truth (north star: mIoU within 1e-3)."""
    from cmda_amd import metrics
    tgt = gpu_target()
    bbc = dict(type='mit_b5', style='pytorch', drop_path_rate=0.1)
    head = dict(type='DAFormerHeadFusion', dropout_ratio=0.1,
                decoder_params=dict(DECODER, train_type='cs2dsec_image+events_together', share_decoder=True), **HEAD)
    model = build_segmentor(dict(type='FusionEncoderDecoder', backbone_image=dict(bbc), backbone_events=dict(bbc),
                                 fusion_module=dict(type='AttentionAvgFusion', in_channels=DIMS, drop_path_rate=0.1),
                                 decode_head=head, train_type='cs2dsec_image+events_together', test_cfg=dict(mode='whole')))
    torch.manual_seed(7)
    model.init_weights()
    ref = oseg.FusionEncoderDecoder(backbone_image=omit.mit_b5(drop_path_rate=0.1), backbone_events=omit.mit_b5(drop_path_rate=0.1),
                                    fusion_module=ofu.AttentionAvgFusion(drop_path_rate=0.1),
                                    decode_head=ohd.DAFormerHeadFusion(dropout_ratio=0.1, share_decoder=True))
    ref.load_state_dict(model.state_dict())
    model.to(tgt.device).eval()
    ref.eval()
    img = seeded_randn((1, 3, 440, 640), 7, 'img')
    ev = seeded_randn((1, 3, 440, 640), 7, 'ev').clamp(-1, 1)
    meta = dict(ori_shape=(440, 640, 3), img_shape=(440, 640, 3), flip=False)
    with torch.no_grad():
        want = ref.encode_decode(img, ev, test_cfg={'output_type': 'fusion'})
    want_pred = torch.softmax(want, 1).argmax(1)[0].numpy()
    gt = labels(1, 640, 7)[0, 0, :440].numpy()
    m_ref = metrics.mean_iou([torch.from_numpy(want_pred)], [torch.from_numpy(gt)], 19, 255)['mIoU'].item()
    for dt, tol, agree_min, miou_tol in ((torch.float32, 1e-3, 0.9995, 1e-3), (torch.bfloat16, 2e-2, 0.985, 1e-3)):   # bf16: 6.7e-3 / 0.9942 / 4.7e-5 measured (three boxes) -- mIoU inside the north star's 1e-3
        rt.set_compute_dtype(dt)
        try:
            got = model.encode_decode(tgt.to(img), tgt.to(ev), test_cfg={'output_type': 'fusion'}).float().cpu()
            pred = model.simple_test(True, warp_image=tgt.to(img), events_vg=tgt.to(ev), img_metas=meta)[0]
        finally:
            rt.set_compute_dtype(torch.float32)
        e = rel(got, want)
        agree = float((pred == want_pred).mean())
        m = metrics.mean_iou([torch.from_numpy(pred)], [torch.from_numpy(gt)], 19, 255)['mIoU'].item()
        print(f'[{dt}] 440x640 fusion simple_test: logits rel err {e:.3e}, label agreement {agree:.5f}, mIoU {m:.5f} vs {m_ref:.5f}')
        assert pred.shape == (440, 640)
        check_le(f'{dt} logits rel err', e, tol, strict=True)
        check_ge(f'{dt} label agreement', agree, agree_min)
        check_le(f'{dt} mIoU abs err', abs(m - m_ref), miou_tol)
    # rescale to a different ori_shape (a test pipeline that resized the image): the second bilinear resize of :926-934
    meta2 = dict(ori_shape=(480, 700, 3), flip=True, flip_direction='horizontal')
    pred2 = model.simple_test(True, warp_image=tgt.to(img), events_vg=tgt.to(ev), img_metas=meta2)[0]
    want2 = torch.softmax(ohd.resize(want, (480, 700)), 1).flip(dims=(3,)).argmax(1)[0].numpy()
    assert pred2.shape == (480, 700)
    check_ge('rescaled + flipped label agreement', float((pred2 == want2).mean()), 0.9995, strict=True)
    # the validation pass as the distributed entry point runs it (parallel.distributed_evaluate: BatchNorm-buffer broadcast, sharded
    # scoring, one histogram exchange; here one rank, no process group): same mIoU as scoring simple_test's label map directly
    from cmda_amd import parallel
    sample = dict(warp_image=tgt.to(img), events_vg=tgt.to(ev), img_metas=meta, gt_semantic_seg=torch.from_numpy(gt))
    res = parallel.distributed_evaluate(model, [sample], 19, 255)
    pred1 = model.simple_test(True, warp_image=tgt.to(img), events_vg=tgt.to(ev), img_metas=meta)[0]
    m1 = metrics.mean_iou([torch.from_numpy(pred1)], [torch.from_numpy(gt)], 19, 255)
    assert abs(res['mIoU'].item() - m1['mIoU'].item()) < 1e-12 and abs(res['aAcc'].item() - m1['aAcc'].item()) < 1e-12
    assert not model.training


@pytest.mark.gpu
def test_checkpoint_roundtrip_gpu(tmp_path):
    """a training checkpoint in the reference's layout ({meta, state_dict with model.* / ema_model.* /
    cyclegan_itrd2en.*, optimizer}) -> function.py:28-37 stripping -> loaded into a FRESH student whose parameters already live in
    a bf16 FlatAdamW store -> identical logits; the optimizer moments survive the round trip."""
    import os
    from cmda_amd import checkpoint as ck, optim
    from cmda_amd.registry import build_train_model
    from test_dacs import SMALL, make_cfg
    tgt = gpu_target()
    rt.set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(3)
        a = build_train_model(make_cfg(SMALL['dims'], SMALL['ch'])).to(tgt.device).train()
        seeded_fill(a.model, 21)
        opt_a = optim.FlatAdamW(a.model, lr=6e-5, weight_decay=0.01)
        a.attach_flat_store(opt_a)
        opt_a.step_count = 5
        opt_a.flat_m.normal_()
        opt_a.flat_v.uniform_()
        f = os.path.join(tmp_path, 'iter_5.pth')
        ck.save_checkpoint(a, f, optimizer=opt_a, meta=dict(iter=5))
        saved = torch.load(f, weights_only=False)
        keys = list(saved['state_dict'])
        assert any(k.startswith('model.') for k in keys) and any(k.startswith('ema_model.') for k in keys)
        assert any(k.startswith('cyclegan_itrd2en.model.') for k in keys)
        released = ck.strip_for_release(saved['state_dict'])
        assert released and all(k.startswith('model.') for k in released)
        g = os.path.join(tmp_path, 'iter_5_state_dict.pth')
        torch.save(dict(released), g)
        b = build_train_model(make_cfg(SMALL['dims'], SMALL['ch'])).to(tgt.device).train()
        seeded_fill(b.model, 22)                          # different weights, then re-homed into the flat bf16 store ...
        opt_b = optim.FlatAdamW(b.model, lr=6e-5, weight_decay=0.01)
        b.attach_flat_store(opt_b)
        ck.load_checkpoint(b, g, strict=False)            # ... and only then loaded: the live bf16 mirrors must follow
        opt_b.load_state_dict(saved['optimizer'])
        assert opt_b.step_count == 5
        for pa, pb in zip(opt_a._order, opt_b._order):   # per parameter: the flat buffers also hold alignment padding
            (la_, ha), (lb_, hb) = opt_a._slot(pa), opt_b._slot(pb)
            assert torch.equal(opt_a.flat_m[la_:ha], opt_b.flat_m[lb_:hb]) and torch.equal(opt_a.flat_v[la_:ha], opt_b.flat_v[lb_:hb])
        img, ev = tgt.to(seeded_randn((1, 3, 64, 64), 3, 'i')), tgt.to(seeded_randn((1, 3, 64, 64), 3, 'e'))
        a.eval(), b.eval()
        la = a.model.encode_decode(img, ev)
        lb = b.model.encode_decode(img, ev)
        assert torch.equal(la, lb), 'logits differ after the checkpoint round trip'
    finally:
        rt.set_compute_dtype(torch.float32)
