"""Image/event feature fusion on the HIP kernels -- registry keys `AttentionAvgFusion`, `AttentionFusion`.

Mirrors mmseg/models/fusion/attention_avg_fusion.py:9-51 and attention_fusion.py:9-59 (ctor kwargs, parameter names
`basic_block.{i}.*` / `linear_block.{i}.*`, `forward(image_features, events_features) -> list of 4 maps`).  Features
travel as (NLC tensor [B*H*W, C], H, W) triples; the reference's flatten/transpose/contiguous round trips vanish.
"""
from functools import partial

import torch
import torch.nn as nn

from . import ops
from . import runtime as rt
from .backbones import Block, Mlp, draw_drop_path
from .registry import FUSION

_LN6 = partial(nn.LayerNorm, eps=1e-6)


@FUSION.register_module()
class AttentionAvgFusion(nn.Module):
    def __init__(self, in_channels=[64, 128, 320, 512], num_heads=1, mlp_ratios=4, qkv_bias=True, qk_scale=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0.05, norm_layer=_LN6, sr_ratios=[8, 4, 2, 1],
                 act_layer=None, init_cfg=None):
        super().__init__()
        self.basic_block = nn.ModuleList([
            Block(dim=in_channels[i // 2], num_heads=num_heads, mlp_ratio=mlp_ratios, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, attn_drop=attn_drop_rate, drop_path=drop_path_rate, norm_layer=norm_layer,
                  sr_ratio=sr_ratios[i // 2]) for i in range(8)])

    def init_weights(self):
        pass  # the reference keeps torch's default initialisation for these blocks (init_cfg=None)

    @ops.sited('fusion')
    def fwd(self, feats_i, feats_e, B, save=True, into=None):
        """into: optional list of 4 pre-allocated tensors the fused maps are written into.
        The image-side blocks (basic_block[0, 2, 4, 6]) and the event-side blocks (1, 3, 5, 7) never see each other's data before
        the average (attention_avg_fusion.py:45-50): with lanes on, the four event-side blocks run on the side lane next to the four
        image-side ones -- the fusion module was ~4.5 ms of small dependent kernels on ONE lane of the step's single-lane phase."""
        outs, saved = [], []
        draw_drop_path(self, list(self.basic_block), B, feats_i[0][0].device)   # one RNG call for the eight blocks
        pools = [blk._dp_pool[0] for blk in self.basic_block if getattr(blk, '_dp_pool', None) is not None]
        ev = []
        with rt.lane('enc', *[x for x, _, _ in feats_e], *pools):
            for i, (xe, H, W) in enumerate(feats_e):
                ev.append(self.basic_block[2 * i + 1].fwd(xe, B, H, W, save=save))
        im = [self.basic_block[2 * i].fwd(xi, B, H, W, save=save) for i, (xi, H, W) in enumerate(feats_i)]
        rt.join_lanes('enc')
        for i, ((yi, si), (ye, se), (_, H, W)) in enumerate(zip(im, ev, feats_i)):
            self.basic_block[2 * i]._dp_pool = self.basic_block[2 * i + 1]._dp_pool = None
            outs.append((ops.axpby(yi, ye, 0.5, 0.5, out=into[i] if into is not None else None), H, W))
            saved.append((si, se, H, W))
        return outs, saved

    @ops.sited('fusion')
    def bwd(self, saved, dfused, B):
        """dfused: list of 4 gradients (or None).  Returns (d image feats, d event feats) as lists."""
        halves = [ops.axpby(d, None, 0.5, 0.0) if d is not None else None for d in dfused]
        de = [None] * len(saved)
        with rt.lane('enc', *[h for h in halves if h is not None]):
            for i, (si, se, H, W) in enumerate(saved):
                if halves[i] is not None:
                    de[i] = self.basic_block[2 * i + 1].bwd(se, halves[i], B, H, W)
            if rt.concurrency():
                ops.gemm_flush_deferred()   # the side lane's own queue of weight gradients (the caller flushes the current lane's)
        di = [self.basic_block[2 * i].bwd(si, halves[i], B, H, W) if halves[i] is not None else None
              for i, (si, se, H, W) in enumerate(saved)]
        rt.join_lanes('enc')
        return di, de

    def forward(self, image_features, events_features):
        from .decode_heads import _HeadBase
        B = image_features[0].shape[0]
        outs, _ = self.fwd(_HeadBase._to_feats(image_features), _HeadBase._to_feats(events_features), B, save=False)
        return [ops.cast(t, torch.float32).view(B, H, W, -1).permute(0, 3, 1, 2) for t, H, W in outs]


@FUSION.register_module()
class AttentionFusion(nn.Module):
    def __init__(self, in_channels=[64, 128, 320, 512], num_heads=1, mlp_ratios=4, qkv_bias=True, qk_scale=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0.05, norm_layer=_LN6, sr_ratios=[8, 4, 2, 1],
                 act_layer=None, init_cfg=None):
        super().__init__()
        self.in_channels = list(in_channels)
        self.basic_block = nn.ModuleList([
            Block(dim=in_channels[i] * 2, num_heads=num_heads, mlp_ratio=mlp_ratios, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, attn_drop=attn_drop_rate, drop_path=drop_path_rate, norm_layer=norm_layer,
                  sr_ratio=sr_ratios[i]) for i in range(4)])
        self.linear_block = nn.ModuleList([
            Mlp(in_features=in_channels[i] * 2, hidden_features=in_channels[i], out_features=in_channels[i])
            for i in range(4)])

    def init_weights(self):
        pass

    @ops.sited('fusion')
    def fwd(self, feats_i, feats_e, B, save=True, into=None):
        outs, saved = [], []
        draw_drop_path(self, list(self.basic_block), B, feats_i[0][0].device)
        for i, ((xi, H, W), (xe, _, _)) in enumerate(zip(feats_i, feats_e)):
            C, M = self.in_channels[i], B * H * W
            cat = torch.empty(M, 2 * C, dtype=rt.compute_dtype(), device=xi.device)
            ops.copy2d(xi, cat, M, C, C, 2 * C)
            ops.copy2d(xe, cat, M, C, C, 2 * C, dst_off=C)
            y, sb = self.basic_block[i].fwd(cat, B, H, W, save=save)
            self.basic_block[i]._dp_pool = None
            z, sm = self.linear_block[i].fwd(y, B, H, W)
            if into is not None:
                z = ops.copy2d(z, into[i], M, C, C, C)
            outs.append((z, H, W))
            saved.append((sb, sm, H, W))
        return outs, saved

    @ops.sited('fusion')
    def bwd(self, saved, dfused, B):
        di, de = [], []
        for i, (sb, sm, H, W) in enumerate(saved):
            d = dfused[i]
            if d is None:
                di.append(None), de.append(None)
                continue
            C, M = self.in_channels[i], B * H * W
            dy = self.linear_block[i].bwd(sm, d, B, H, W)
            dcat = self.basic_block[i].bwd(sb, dy, B, H, W)
            a = torch.empty(M, C, dtype=rt.compute_dtype(), device=d.device)
            b = torch.empty(M, C, dtype=rt.compute_dtype(), device=d.device)
            ops.copy2d(dcat, a, M, C, 2 * C, C)
            ops.copy2d(dcat, b, M, C, 2 * C, C, src_off=C)
            di.append(a), de.append(b)
        return di, de

    def forward(self, image_features, events_features):
        from .decode_heads import _HeadBase
        B = image_features[0].shape[0]
        outs, _ = self.fwd(_HeadBase._to_feats(image_features), _HeadBase._to_feats(events_features), B, save=False)
        return [ops.cast(t, torch.float32).view(B, H, W, -1).permute(0, 3, 1, 2) for t, H, W in outs]
