"""ORACLE (test infrastructure, not product code): CPU fp32 restatement of the MiT (SegFormer) encoder.

Follows the reference algorithm of mmseg/models/backbones/mix_transformer.py:
  Mlp :20-44, Attention :47-105, Block :108-148, OverlapPatchEmbed :151-183,
  MixVisionTransformer :186-440 (forward_features :397-433, _init_weights :324-337), DWConv :443-455, mit_b5 :538-551.
Parameter names match the reference so that its state_dicts load here unchanged.

Pinning: tests/golden/*.npz hold outputs of the *reference's own modules* (imported unmodified in the authoring
container by tests/golden/make_golden.py) for seeded weights/inputs; tests/test_oracle_golden.py checks this file
against them.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import oracle/.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def drop_path(x, rate, training, generator=None):
    """timm 0.3.2 DropPath: per-sample keep mask, scaled by 1/keep."""
    if rate == 0.0 or not training:
        return x
    keep = 1.0 - rate
    mask = keep + torch.rand((x.shape[0],) + (1,) * (x.dim() - 1), dtype=x.dtype, device=x.device, generator=generator)
    return x.div(keep) * mask.floor_()


class DropPath(nn.Module):
    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        return drop_path(x, self.drop_prob, self.training)


class DWConv(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)

    def forward(self, x, H, W):
        B, N, C = x.shape
        y = self.dwconv(x.transpose(1, 2).reshape(B, C, H, W))
        return y.flatten(2).transpose(1, 2)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.dwconv = DWConv(hidden_features)
        self.fc2 = nn.Linear(hidden_features, out_features)

    def forward(self, x, H, W):
        return self.fc2(F.gelu(self.dwconv(self.fc1(x), H, W)))


class Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias, sr_ratio):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.sr_ratio = sr_ratio
        if sr_ratio > 1:
            self.sr = nn.Conv2d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)
            self.norm = nn.LayerNorm(dim)  # eps 1e-5

    def forward(self, x, H, W):
        B, N, C = x.shape
        h, d = self.num_heads, C // self.num_heads
        q = self.q(x).reshape(B, N, h, d).transpose(1, 2)
        if self.sr_ratio > 1:
            xs = self.sr(x.transpose(1, 2).reshape(B, C, H, W)).reshape(B, C, -1).transpose(1, 2)
            xs = self.norm(xs)
        else:
            xs = x
        kv = self.kv(xs).reshape(B, -1, 2, h, d).permute(2, 0, 3, 1, 4)
        k, v = kv[0], kv[1]
        attn = ((q @ k.transpose(-2, -1)) * self.scale).softmax(dim=-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(B, N, C))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, drop_path=0.0, sr_ratio=1, eps=1e-6):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, num_heads, qkv_bias, sr_ratio)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x, H, W):
        x = x + self.drop_path(self.attn(self.norm1(x), H, W))
        return x + self.drop_path(self.mlp(self.norm2(x), H, W))


class OverlapPatchEmbed(nn.Module):
    def __init__(self, patch_size, stride, in_chans, embed_dim):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, patch_size, stride, patch_size // 2)
        self.norm = nn.LayerNorm(embed_dim)  # eps 1e-5

    def forward(self, x):
        x = self.proj(x)
        H, W = x.shape[2:]
        return self.norm(x.flatten(2).transpose(1, 2)), H, W


class MixVisionTransformer(nn.Module):
    def __init__(self, in_chans=3, embed_dims=(64, 128, 320, 512), num_heads=(1, 2, 5, 8), mlp_ratios=(4, 4, 4, 4),
                 qkv_bias=True, drop_path_rate=0.1, depths=(3, 6, 40, 3), sr_ratios=(8, 4, 2, 1), eps=1e-6, **_):
        super().__init__()
        self.depths = depths
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        cur = 0
        for s in range(4):
            cin = in_chans if s == 0 else embed_dims[s - 1]
            setattr(self, f'patch_embed{s + 1}', OverlapPatchEmbed(7 if s == 0 else 3, 4 if s == 0 else 2, cin, embed_dims[s]))
            setattr(self, f'block{s + 1}', nn.ModuleList([
                Block(embed_dims[s], num_heads[s], mlp_ratios[s], qkv_bias, dpr[cur + i], sr_ratios[s], eps)
                for i in range(depths[s])]))
            setattr(self, f'norm{s + 1}', nn.LayerNorm(embed_dims[s], eps=eps))
            cur += depths[s]

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Conv2d):
                fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
                m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
                if m.bias is not None:
                    m.bias.data.zero_()

    def forward(self, x):
        B = x.shape[0]
        outs = []
        for s in range(1, 5):
            x, H, W = getattr(self, f'patch_embed{s}')(x)
            for blk in getattr(self, f'block{s}'):
                x = blk(x, H, W)
            x = getattr(self, f'norm{s}')(x)
            x = x.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()
            outs.append(x)
        return outs


def mit_b5(**kw):
    return MixVisionTransformer(embed_dims=(64, 128, 320, 512), num_heads=(1, 2, 5, 8), depths=(3, 6, 40, 3),
                                sr_ratios=(8, 4, 2, 1), qkv_bias=True, **kw)
