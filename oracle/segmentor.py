"""ORACLE (test infrastructure): the fusion segmentor.

Follows mmseg/models/segmentors/encoder_decoder.py FusionEncoderDecoder :625-1003
(extract_feat :698-721, encode_decode :723-746, forward_train :794-831) for the two train types of configs/fusion/*.
Pinned by tests/golden.
"""
import torch.nn as nn

from .fusion import AttentionAvgFusion, AttentionFusion
from .head import DAFormerHead, DAFormerHeadFusion, resize
from .mit import mit_b5


class EncoderDecoder(nn.Module):
    """Single-modality MiT-B5 + DAFormerHead (BASELINE.json configs[0]/[1])."""

    def __init__(self, backbone=None, decode_head=None):
        super().__init__()
        self.backbone = backbone if backbone is not None else mit_b5()
        self.decode_head = decode_head if decode_head is not None else DAFormerHead()

    def encode_decode(self, img):
        return resize(self.decode_head(self.backbone(img)), img.shape[2:])

    def forward_train(self, img, gt, seg_weight=None):
        losses, logits = self.decode_head.forward_train(self.backbone(img), gt, seg_weight)
        return {'decode.' + k: v for k, v in losses.items()}, logits


class FusionEncoderDecoder(nn.Module):
    def __init__(self, backbone_image=None, backbone_events=None, fusion_module=None, decode_head=None,
                 fusion_isr_module=None, train_type='cs2dsec_image+events_together'):
        super().__init__()
        self.train_type = train_type
        self.backbone_image = backbone_image if backbone_image is not None else mit_b5()
        self.backbone_events = backbone_events if backbone_events is not None else mit_b5()
        self.fusion_module = fusion_module if fusion_module is not None else AttentionAvgFusion()
        if fusion_isr_module is not None:
            self.fusion_isr_module = fusion_isr_module
        self.decode_head = decode_head if decode_head is not None else DAFormerHeadFusion()

    def extract_feat(self, image, events, img_self_res=None, cfg=None):
        cfg = cfg or {}
        f_image = self.backbone_image(image.detach()) if image is not None else None
        f_events = self.backbone_events(events.detach()) if events is not None else None
        f_isr = self.backbone_events(img_self_res.detach()) if img_self_res is not None else None
        if cfg.get('no_fusion'):
            f_fusion = None
        elif cfg.get('fusion_isr'):
            f_fusion = self.fusion_isr_module(f_image, f_events if img_self_res is None else f_isr)
        elif cfg.get('fusion_all'):
            a, b = self.fusion_isr_module(f_image, f_isr), self.fusion_module(f_image, f_events)
            f_fusion = [(x + y) / 2 for x, y in zip(a, b)]
        else:
            f_fusion = self.fusion_module(f_image, f_events) if (self.fusion_module is not None and events is not None) else None
        return {'f_image': f_image, 'f_events': f_events, 'f_fusion': f_fusion, 'f_img_self_res': f_isr}

    def encode_decode(self, img, events, img_self_res=None, output_features=False, test_cfg=None):
        test_cfg = test_cfg or {'output_type': 'fusion'}
        x = self.extract_feat(img, events, img_self_res, cfg=test_cfg)
        if events is None:
            test_cfg = {'output_type': 'image'}
        out = self.decode_head(x, test_cfg)
        size = img.shape[2:] if img is not None else events.shape[2:]
        if output_features:
            return {k: (resize(v, size) if v is not None else None) for k, v in out.items()}
        return resize(out[test_cfg['output_type'] + '_output'], size)

    def forward_train(self, inputs, gt, seg_weight=None, return_feat=False, cfg=None):
        x = self.extract_feat(inputs['image'], inputs['events'], inputs.get('img_self_res'), cfg=cfg)
        losses = {}
        if return_feat:
            losses['features'] = x
        loss_decode, pred = self.decode_head.forward_train(x, gt, seg_weight, cfg)
        losses.update({'decode.' + k: v for k, v in loss_decode.items()})
        return losses, pred
