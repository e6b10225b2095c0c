"""ORACLE (test infrastructure): image/event feature fusion modules.

Follows mmseg/models/fusion/attention_avg_fusion.py:9-51 (AttentionAvgFusion) and
mmseg/models/fusion/attention_fusion.py:9-59 (AttentionFusion).  Pinned by tests/golden.
"""
import torch
import torch.nn as nn

from .mit import Block, Mlp


class AttentionAvgFusion(nn.Module):
    def __init__(self, in_channels=(64, 128, 320, 512), num_heads=1, mlp_ratios=4, qkv_bias=True,
                 drop_path_rate=0.05, sr_ratios=(8, 4, 2, 1), **_):
        super().__init__()
        self.basic_block = nn.ModuleList([
            Block(in_channels[i // 2], num_heads, mlp_ratios, qkv_bias, drop_path_rate, sr_ratios[i // 2]) for i in range(8)])

    def forward(self, image_features, events_features):
        outs = []
        for i, (fi, fe) in enumerate(zip(image_features, events_features)):
            B, _, H, W = fi.shape
            xi = self.basic_block[2 * i](fi.flatten(2).transpose(1, 2), H, W)
            xe = self.basic_block[2 * i + 1](fe.flatten(2).transpose(1, 2), H, W)
            outs.append(((xi + xe) / 2).reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous())
        return outs


class AttentionFusion(nn.Module):
    def __init__(self, in_channels=(64, 128, 320, 512), num_heads=1, mlp_ratios=4, qkv_bias=True,
                 drop_path_rate=0.05, sr_ratios=(8, 4, 2, 1), **_):
        super().__init__()
        self.basic_block = nn.ModuleList([
            Block(in_channels[i] * 2, num_heads, mlp_ratios, qkv_bias, drop_path_rate, sr_ratios[i]) for i in range(4)])
        self.linear_block = nn.ModuleList([Mlp(in_channels[i] * 2, in_channels[i], in_channels[i]) for i in range(4)])

    def forward(self, image_features, events_features):
        outs = []
        for i, (fi, fe) in enumerate(zip(image_features, events_features)):
            x = torch.cat((fi, fe), dim=1)
            B, _, H, W = x.shape
            x = self.basic_block[i](x.flatten(2).transpose(1, 2), H, W)
            x = self.linear_block[i](x, H, W)
            outs.append(x.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous())
        return outs
