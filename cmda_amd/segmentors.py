"""Segmentors on the HIP kernels -- registry keys `EncoderDecoder`, `FusionEncoderDecoder`.

Mirrors mmseg/models/segmentors/encoder_decoder.py (EncoderDecoder :19-300; FusionEncoderDecoder :625-1003:
extract_feat :698-721, encode_decode :723-746, forward_train :794-831) and base.py `_parse_losses` :710-743.
The whole student pass (backbones -> fusion -> decode head -> fused up-sample + CE) is scheduled by hand on the
kernel library; autograd sees one node per forward_train (`_TrainFn`) whose backward replays the hand-written
backward pass and accumulates parameter gradients in place.
"""
from collections import OrderedDict

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
        dfs = self.decode_head.bwd_train(sv_h, B, gscale)
        self.backbone.bwd(sv_b, [dfs.get(i) for i in range(4)])

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
