"""Segmentors on the HIP kernels -- registry keys `EncoderDecoder`, `FusionEncoderDecoder`.

Mirrors mmseg/models/segmentors/encoder_decoder.py (EncoderDecoder :19-300; FusionEncoderDecoder :625-1003:
extract_feat :698-721, encode_decode :723-746, forward_train :794-831) and base.py `_parse_losses` :710-743.
The whole student pass (backbones -> fusion -> decode head -> fused up-sample + CE) is scheduled by hand on the
kernel library; autograd sees one node per forward_train (`_TrainFn`) whose backward replays the hand-written
backward pass and accumulates parameter gradients in place.
"""
from collections import OrderedDict

import os

import torch
import torch.nn as nn

from . import ops
from . import runtime as rt
from .registry import SEGMENTORS, build_backbone, build_fusion, build_head


def add_prefix(d, prefix):
    return {f'{prefix}.{k}': v for k, v in d.items()}


def parse_losses(losses):
    """base.py:710-743 without the host syncs: returns (loss tensor, log_vars of *device* scalars)."""
    log_vars = OrderedDict()
    for name, value in losses.items():
        if isinstance(value, torch.Tensor):
            log_vars[name] = value.mean()
        elif isinstance(value, list):
            log_vars[name] = sum(v.mean() for v in value)
        else:
            raise TypeError(f'{name} is not a tensor or list of tensors')
    loss = sum(v for k, v in log_vars.items() if 'loss' in k)
    log_vars['loss'] = loss
    return loss, log_vars


class _TrainFn(torch.autograd.Function):
    """loss = runner.train_fwd(...); backward(dloss) -> runner.train_bwd(saved, dloss)."""

    @staticmethod
    def forward(ctx, runner, anchor, args):
        loss, aux, saved = runner.train_fwd(*args)
        ctx.runner, ctx.saved = runner, saved
        ctx.aux = aux
        return loss

    @staticmethod
    def backward(ctx, dloss):
        ctx.runner.train_bwd(ctx.saved, dloss.contiguous().float().view(1))
        ctx.saved = None
        return None, None, None


@SEGMENTORS.register_module()
class EncoderDecoder(nn.Module):
    """Single-modality MiT + DAFormerHead (BASELINE.json configs[0]/[1])."""

    def __init__(self, backbone, decode_head, neck=None, auxiliary_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None, init_cfg=None):
        super().__init__()
        assert neck is None and auxiliary_head is None
        if pretrained is not None:
            assert backbone.get('pretrained') is None, 'both backbone and segmentor set pretrained weight'
            backbone = dict(backbone, pretrained=pretrained)
        self.backbone = build_backbone(backbone)
        self.decode_head = build_head(decode_head)
        self.align_corners = self.decode_head.align_corners
        self.num_classes = self.decode_head.num_classes
        self.train_cfg, self.test_cfg = train_cfg, test_cfg

    def init_weights(self):
        self.backbone.init_weights()
        self.decode_head.init_weights()

    def extract_feat(self, img):
        return self.backbone(img)

    # hand-scheduled training pass
    def train_fwd(self, img, gt, seg_weight):
        B = img.shape[0]
        feats, sv_b = self.backbone.fwd(img)
        losses, logits, sv_h = self.decode_head.fwd_train(feats, B, gt, seg_weight)
        return losses['loss_seg'], (losses, logits), (sv_b, sv_h, B)

    def train_bwd(self, saved, gscale):
        sv_b, sv_h, B = saved
        with ops.ln_deferral():   # LayerNorm parameter gradients of the whole pass folded by one launch at the end
            dfs = self.decode_head.bwd_train(sv_h, B, gscale)
            rt.notify_grads_ready('decode_head', self.decode_head)
            self.backbone.bwd(sv_b, [dfs.get(i) for i in range(4)])
        rt.join_lanes('wgrad')

    def forward_train(self, img, img_metas=None, gt_semantic_seg=None, seg_weight=None, return_feat=False):
        holder = {}
        loss = _TrainFn.apply(_Capture(self, holder), rt.anchor(img.device), (img, gt_semantic_seg, seg_weight))
        losses, logits = holder['aux']
        out = add_prefix({'loss_seg': loss, 'acc_seg': losses['acc_seg']}, 'decode')
        return out, logits.permute(0, 3, 1, 2)

    def encode_decode(self, img, img_metas=None):
        B, _, H, W = img.shape
        with torch.no_grad():
            feats, _ = self.backbone.fwd(img, save=False)
            logits, _ = self.decode_head.fwd(feats, B)
            return ops.upsample_logits_nchw(logits, H, W)

    def simple_test(self, img, img_meta=None, rescale=True):
        """encoder_decoder.py:222-285 with test_cfg mode 'whole': logits at the input size, resized to img_meta['ori_shape'] when
        `rescale`, flipped back when the test pipeline flipped; per-image label maps (numpy) -- the reference's soft-max before
        the argmax is monotone and skipped."""
        seg_logit = self.encode_decode(img, img_meta)
        meta = (img_meta[0] if isinstance(img_meta, (list, tuple)) else img_meta) or {}
        if rescale and meta.get('ori_shape') is not None and tuple(meta['ori_shape'][:2]) != tuple(seg_logit.shape[2:]):
            h, w = meta['ori_shape'][:2]
            seg_logit = ops.upsample_logits_nchw(seg_logit.permute(0, 2, 3, 1).contiguous(), h, w)
        if meta.get('flip'):
            seg_logit = seg_logit.flip(dims=(3,) if meta['flip_direction'] == 'horizontal' else (2,))
        return list(seg_logit.argmax(dim=1).cpu().numpy())


class _Capture:
    """Adapter so that _TrainFn can hand the auxiliary outputs (logits, accuracy) back to forward_train."""

    def __init__(self, model, holder):
        self.model, self.holder = model, holder

    def train_fwd(self, *args):
        loss, aux, saved = self.model.train_fwd(*args)
        self.holder['aux'] = aux
        return loss, aux, saved

    def train_bwd(self, saved, gscale):
        return self.model.train_bwd(saved, gscale)


def _sum_grads(a, b):
    """element-wise sum of two per-level gradient lists/dicts (entries may be None)."""
    out = []
    for i in range(4):
        x = a[i] if a is not None else None
        y = b[i] if b is not None else None
        if isinstance(a, dict):
            x = a.get(i)
        if isinstance(b, dict):
            y = b.get(i)
        out.append(y if x is None else (x if y is None else ops.axpby(x, y, 1.0, 1.0)))
    return out


@SEGMENTORS.register_module()
class FusionEncoderDecoder(nn.Module):
    """encoder_decoder.py:625-1003 for train types 'cs2dsec_image+events_together' / 'cs2dsec_image+events' /
    'cs2dz_image+raw-isr' (the ones configs/fusion/* use)."""

    TRAIN_TYPES = {'cs2dsec_image+events', 'cs2dz_image+d2n-isr', 'cs2dz_image+raw-isr', 'cs2dz_image+raw-isr_no-fusion',
                   'cs2dz_image+raw-isr_split', 'cs2dsec_image+events_together'}

    def __init__(self, backbone_image, backbone_events, fusion_module, decode_head, neck=None, auxiliary_head=None,
                 train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None, **kwargs):
        super().__init__()
        assert kwargs['train_type'] in self.TRAIN_TYPES
        self.train_type = kwargs['train_type']
        assert neck is None and auxiliary_head is None
        if pretrained is not None:
            assert backbone_events.get('pretrained') is None and backbone_image.get('pretrained') is None, \
                'both backbone and segmentor set pretrained weight'
            backbone_events = dict(backbone_events, pretrained=pretrained)
            backbone_image = dict(backbone_image, pretrained=pretrained)
        self.backbone_image = build_backbone(backbone_image)
        self.backbone_events = build_backbone(backbone_events)
        if self.train_type in {'cs2dsec_image+events', 'cs2dz_image+raw-isr', 'cs2dsec_image+events_together'}:
            self.fusion_module = build_fusion(fusion_module)
            fim = kwargs.get('fusion_isr_module')
            if fim is not None and fim.get('type', '') != '':
                self.fusion_isr_module = build_fusion(fim)
        else:
            self.fusion_module = None
        self.decode_head = build_head(decode_head)
        self.align_corners = self.decode_head.align_corners
        self.num_classes = self.decode_head.num_classes
        self.train_cfg, self.test_cfg = train_cfg, test_cfg

    def init_weights(self):
        for m in (self.backbone_image, self.backbone_events, self.decode_head):
            m.init_weights()

    # -- feature extraction (extract_feat :698-721) --------------------------------------------------------------
    def _extract(self, image, events, img_self_res, cfg, save):
        cfg = cfg or {}
        B = (image if image is not None else events).shape[0]
        sv = {}
        f_image = f_events = f_isr = None
        if image is not None:
            f_image, sv['image'] = self.backbone_image.fwd(image, save=save)
        if events is not None:
            f_events, sv['events'] = self.backbone_events.fwd(events, save=save)
        if img_self_res is not None:
            f_isr, sv['isr'] = self.backbone_events.fwd(img_self_res, save=save)
        f_fusion = None
        if cfg.get('no_fusion'):
            pass
        elif cfg.get('fusion_isr'):
            second = 'events' if img_self_res is None else 'isr'
            f_fusion, sv['fusion_isr'] = self.fusion_isr_module.fwd(f_image, f_events if second == 'events' else f_isr, B, save)
            sv['fusion_isr_second'] = second
        elif cfg.get('fusion_all'):
            a, sv['fusion_isr'] = self.fusion_isr_module.fwd(f_image, f_isr, B, save)
            sv['fusion_isr_second'] = 'isr'
            b, sv['fusion'] = self.fusion_module.fwd(f_image, f_events, B, save)
            f_fusion = [(ops.axpby(x[0], y[0], 0.5, 0.5), x[1], x[2]) for x, y in zip(a, b)]
            sv['fusion_all'] = True
        elif self.fusion_module is not None and events is not None:
            f_fusion, sv['fusion'] = self.fusion_module.fwd(f_image, f_events, B, save)
        feats = {'f_image': f_image, 'f_events': f_events, 'f_fusion': f_fusion, 'f_img_self_res': f_isr}
        return feats, sv, B

    def _extract_bwd(self, sv, dfeats, B):
        d_img = dfeats.get('f_image')
        d_evt = dfeats.get('f_events')
        d_isr = dfeats.get('f_img_self_res')
        d_fus = dfeats.get('f_fusion')
        if d_fus is not None:
            d_fus = [d_fus.get(i) for i in range(4)]
            if sv.get('fusion_all'):
                d_fus = [ops.axpby(d, None, 0.5, 0.0) if d is not None else None for d in d_fus]
            if 'fusion' in sv:
                di, de = self.fusion_module.bwd(sv['fusion'], d_fus, B)
                d_img, d_evt = _sum_grads(d_img, di), _sum_grads(d_evt, de)
            if 'fusion_isr' in sv:
                di, de = self.fusion_isr_module.bwd(sv['fusion_isr'], d_fus, B)
                d_img = _sum_grads(d_img, di)
                if sv['fusion_isr_second'] == 'isr':
                    d_isr = _sum_grads(d_isr, de)
                else:
                    d_evt = _sum_grads(d_evt, de)
        as_list = lambda d: [d.get(i) for i in range(4)] if isinstance(d, dict) else d
        if 'image' in sv and d_img is not None:
            self.backbone_image.bwd(sv['image'], as_list(d_img))
        if 'isr' in sv and d_isr is not None:
            self.backbone_events.bwd(sv['isr'], as_list(d_isr))
        if 'events' in sv and d_evt is not None:
            self.backbone_events.bwd(sv['events'], as_list(d_evt))

    # -- joint pass: events + ISR through the event encoder as ONE batch, all feature sets through the shared decoder at once ----
    def _joint_ok(self, image, events, cfg):
        cfg = cfg or {}
        return (image is not None and events is not None and self.fusion_module is not None
                and not (cfg.get('no_fusion') or cfg.get('fusion_isr') or cfg.get('fusion_all'))
                and hasattr(self.decode_head, 'joint_ok') and self.decode_head.joint_ok()
                and getattr(self, 'joint_passes', True))

    def _extract_joint(self, image, events, img_self_res, save):
        """extract_feat (:698-721) for the default fusion route with the outputs laid out for DAFormerHeadFusion.fwd_joint: per
        level one buffer J_l [G*B*N_l, C_l] holding [image | fusion | events | ISR] blocks.  The image encoder writes block 0,
        the event encoder -- run ONCE over the events and the ISR as a 2B batch (same weights, :703-712) -- blocks 2 and 3, the
        fusion module block 1.  No feature is copied.  Every input may be a LIST of P tensors (the samples of P passes that run
        the same weights, `train_fwd_passes`): B below is then the total over the passes and each block holds pass 0's samples,
        then pass 1's, ..."""
        as_list = lambda t: list(t) if isinstance(t, (list, tuple)) else [t]
        image, events = as_list(image), as_list(events)
        img_self_res = as_list(img_self_res) if img_self_res is not None else None
        B = sum(t.shape[0] for t in image)
        H, W = image[0].shape[2:]
        names = ('image', 'fusion', 'events') + (('isr',) if img_self_res is not None else ())
        G = len(names)
        dims = self.backbone_image.embed_dims
        shapes = self.backbone_image.feature_shapes(H, W)
        dev = image[0].device
        joint = [torch.empty(G * B * h * w, c, dtype=rt.compute_dtype(), device=dev) for (h, w), c in zip(shapes, dims)]
        n = [B * h * w for h, w in shapes]
        ev_in = events + (img_self_res or [])
        with rt.lane('enc', *ev_in, *joint):   # the two encoders are independent until the fusion module: side by side
            f_ev, sv_e = self.backbone_events.fwd(ev_in, save=save, out_feats=[J[2 * m:G * m] for J, m in zip(joint, n)])
        f_image, sv_i = self.backbone_image.fwd(image, save=save, out_feats=[J[:m] for J, m in zip(joint, n)])
        rt.join_lanes('enc')
        f_events = [(J[2 * m:3 * m], h, w) for J, m, (h, w) in zip(joint, n, shapes)]
        _, sv_f = self.fusion_module.fwd(f_image, f_events, B, save, into=[J[m:2 * m] for J, m in zip(joint, n)])
        feats = [(J, h, w) for J, (h, w) in zip(joint, shapes)]
        return feats, names, (sv_i, sv_e, sv_f, n, G), B

    def _extract_joint_bwd(self, sv, dJ, B, tail_key=None):
        """dJ: {level: d J_l}; the gradient blocks are consumed in place (fusion contributions are added into blocks 0 and 2)"""
        sv_i, sv_e, sv_f, n, G = sv
        d = [dJ.get(i) for i in range(4)]
        d_fus = [(t[m:2 * m] if t is not None else None) for t, m in zip(d, n)]
        di, de = self.fusion_module.bwd(sv_f, d_fus, B)
        ops.gemm_flush_deferred()       # the fusion blocks' queued weight gradients
        d_img, d_ev = [], []
        for t, m, a, b in zip(d, n, di, de):
            if t is None:
                d_img.append(a)
                assert b is None, 'event-encoder gradient without a joint gradient buffer'
                d_ev.append(None)
                continue
            if a is not None:
                ops.axpby(t[:m], a, 1.0, 1.0, out=t[:m])
            if b is not None:
                ops.axpby(t[2 * m:3 * m], b, 1.0, 1.0, out=t[2 * m:3 * m])
            d_img.append(t[:m])
            d_ev.append(t[2 * m:G * m])
        with rt.lane('enc', *[t for t in d if t is not None]):
            self.backbone_events.bwd(sv_e, d_ev)
        self.backbone_image.bwd(sv_i, d_img)
        if tail_key is not None:   # the decode head's postponed weight gradients: behind the (shorter) image-encoder chain
            ops.run_tail(tail_key)
        rt.join_lanes('enc')
        rt.join_lanes('wgrad')

    # -- hand-scheduled training pass ---------------------------------------------------------------------------------
    def train_fwd(self, inputs, gt, seg_weight, cfg):
        if self._joint_ok(inputs['image'], inputs['events'], cfg):
            feats, names, sv, B = self._extract_joint(inputs['image'], inputs['events'], inputs.get('img_self_res'), True)
            losses, logits, sv_h = self.decode_head.fwd_train_joint(feats, names, B, gt, seg_weight, cfg)
            return losses['loss_seg'], (losses, logits, feats), ('joint', sv, sv_h, B)
        feats, sv, B = self._extract(inputs['image'], inputs['events'], inputs.get('img_self_res'), cfg, True)
        losses, logits, sv_h = self.decode_head.fwd_train(feats, B, gt, seg_weight, cfg)
        return losses['loss_seg'], (losses, logits, feats), (sv, sv_h, B)

    def train_fwd_passes(self, passes, cfg, before_head=None):
        """P training passes with the same weights as ONE pass over P*B samples (passes: list of (inputs, gt, seg_weight)): the
        source and the mixed step of a DACS iteration (dacs.py:489-523, :820-860) differ only in their inputs and targets, and
        the reference's `backward()` calls just add their gradients up.  Per-pass state -- BatchNorm batch statistics and the
        order of the running-statistic updates, the loss normalisation -- is kept per pass (DAFormerHeadFusion.fwd_joint).
        Returns ([(loss, losses)] per pass, saved); `train_bwd(saved, gscale)` back-propagates the SUM of the pass losses.
        before_head: called between the encoders / fusion blocks and the decode head -- the first use of the targets."""
        first = passes[0][0]
        assert self._joint_ok(first['image'], first['events'], cfg)
        P = len(passes)
        isr = [p[0]['img_self_res'] for p in passes] if first.get('img_self_res') is not None else None
        feats, names, sv, Bt = self._extract_joint([p[0]['image'] for p in passes], [p[0]['events'] for p in passes], isr, True)
        if before_head is not None:   # the targets (gt, seg_weight) may come from another lane: joined here, behind the encoders (uda.DACS)
            before_head()
        losses, logits, sv_h = self.decode_head.fwd_train_joint(feats, names, Bt // P, [p[1] for p in passes],
                                                                [p[2] for p in passes], cfg, passes=P)
        return [(l['loss_seg'], l) for l in losses], ('joint', sv, sv_h, Bt, P)

    def train_bwd(self, saved, gscale):
        with ops.ln_deferral():   # LayerNorm parameter gradients of the whole pass folded by one launch at the end
            if saved[0] == 'joint':
                _, sv, sv_h, B = saved[:4]
                P = saved[4] if len(saved) > 4 else 1
                # TAIL mode (single GPU, lanes on): the decode head's weight gradients (dense GEMMs + the depthwise weight-gradient
                # stencils, ~4 ms at 2 + 2 samples) are off the critical path.  The image encoder's backward chain ends ~2 ms before
                # the event encoder's (half the samples), so they are queued under their own key and launched on the main lane
                # BEHIND the image encoder's backward pass, in the gap before the join.  (With a gradient exchange armed the head's
                # gradients must be final before the encoders start: flushed in place, as before.)
                tail = (rt.concurrency() and rt.grad_ready_hook is None and not rt.lane_enabled('hw') and
                        os.environ.get('CMDA_HEAD_TAIL', '1') != '0')
                if tail:
                    ops.GD_QUEUE_KEY = 'main/headtail'
                try:
                    dJ = self.decode_head.bwd_train_joint(sv_h, B // P, gscale)
                finally:
                    ops.GD_QUEUE_KEY = None
                if tail:
                    self._extract_joint_bwd(sv, dJ, B, tail_key='main/headtail')
                    return
                if rt.lane_enabled('hw') and rt.grad_ready_hook is None:
                    # the decode head's queued weight gradients (dense 256 x 256-tile GEMMs, ~3.5 ms at 2 + 2 samples) are off the
                    # critical path: a third queue runs them underneath the encoders' latency-bound backward chains
                    with rt.lane('hw', *ops.gemm_deferred_tensors()):
                        ops.gemm_flush_deferred(from_lane='main')
                else:
                    ops.gemm_flush_deferred()   # the decode head's queued weight gradients
                rt.notify_grads_ready('decode_head', self.decode_head)
                self._extract_joint_bwd(sv, dJ, B)
                rt.join_lanes('hw')
                return
            sv, sv_h, B = saved
            dfeats = self.decode_head.bwd_train(sv_h, B, gscale)
            rt.notify_grads_ready('decode_head', self.decode_head)
            self._extract_bwd(sv, dfeats, B)
            rt.join_lanes('wgrad')

    def forward_train(self, inputs, gt_semantic_seg, seg_weight=None, return_feat=False, cfg=None):
        holder = {}
        dev = inputs['image'].device
        loss = _TrainFn.apply(_Capture(self, holder), rt.anchor(dev), (inputs, gt_semantic_seg, seg_weight, cfg))
        losses, logits, feats = holder['aux']
        out = {}
        if return_feat:
            out['features'] = feats
        out.update(add_prefix({'loss_seg': loss, 'acc_seg': losses['acc_seg']}, 'decode'))
        pred = {k: (v.permute(0, 3, 1, 2) if v is not None else None) for k, v in logits.items()}
        return out, pred

    # -- inference / teacher ---------------------------------------------------------------------------------------------
    def encode_decode_lowres(self, img, events, img_self_res=None, test_cfg=None):
        """dict of fp32 NHWC logits at 1/4 resolution (the fused kernels up-sample on the fly)."""
        with torch.no_grad():
            if self._joint_ok(img, events, test_cfg):
                feats, names, _, B = self._extract_joint(img, events, img_self_res, False)
                out, _ = self.decode_head.fwd_joint(feats, names, B)
                return out
            feats, _, B = self._extract(img, events, img_self_res, test_cfg, False)
            out, _ = self.decode_head.fwd(feats, B)
        return out

    def encode_decode(self, img, events, img_self_res=None, output_features=False, test_cfg={'output_type': 'fusion'}):
        out = self.encode_decode_lowres(img, events, img_self_res, test_cfg)
        if events is None:
            test_cfg = {'output_type': 'image'}
        H, W = (img if img is not None else events).shape[2:]
        if output_features:
            return {k: (ops.upsample_logits_nchw(v, H, W) if v is not None else None) for k, v in out.items()}
        return ops.upsample_logits_nchw(out[test_cfg['output_type'] + '_output'], H, W)

    def whole_inference(self, rescale, **kwargs):
        """encoder_decoder.py:897-936: logits at the input size, resized once more to img_metas['ori_shape'] when `rescale`"""
        img = kwargs['warp_image'] if 'warp_image' in kwargs else kwargs['image']
        test_cfg = kwargs.get('test_cfg') or {'output_type': 'fusion'}
        if self.train_type in {'cs2dsec_image+events', 'cs2dsec_image+events_together'} and 'events_vg' in kwargs:
            events = kwargs['events_vg']
        elif self.train_type == 'cs2dz_image+raw-isr' and test_cfg['output_type'] == 'image_isr':
            events = kwargs['night_isr']
        else:
            events = None
        if self.train_type == 'cs2dz_image+raw-isr':
            test_cfg = {'output_type': 'fusion'} if test_cfg['output_type'] == 'image_isr' else {'output_type': 'image'}
        seg_logit = self.encode_decode(img, events, test_cfg=test_cfg)
        meta = kwargs.get('img_metas')
        if rescale and meta is not None:
            meta = meta[0] if isinstance(meta, (list, tuple)) else meta
            size = tuple(meta['ori_shape'][:2])
            if size != tuple(seg_logit.shape[2:]):
                seg_logit = ops.upsample_logits_nchw(seg_logit.permute(0, 2, 3, 1).contiguous(), size[0], size[1])
        return seg_logit

    def inference(self, rescale, **kwargs):
        """encoder_decoder.py:938-971 (test_cfg.mode 'whole'): soft-max of the logits, flipped back when the test pipeline flipped"""
        seg_logit = self.whole_inference(rescale, **kwargs)
        output = torch.softmax(seg_logit, dim=1)
        meta = kwargs.get('img_metas')
        meta = (meta[0] if isinstance(meta, (list, tuple)) else meta) or {}
        if meta.get('flip'):
            assert meta['flip_direction'] in ('horizontal', 'vertical')
            output = output.flip(dims=(3,) if meta['flip_direction'] == 'horizontal' else (2,))
        return output

    def simple_test(self, rescale=True, **kwargs):
        """encoder_decoder.py:973-984: per-image label maps (numpy)"""
        return list(self.inference(rescale, **kwargs).argmax(dim=1).cpu().numpy())
