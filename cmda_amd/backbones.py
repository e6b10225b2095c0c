"""MiT (SegFormer) encoders on the HIP kernels -- registry keys `mit_b0` .. `mit_b5`, `MixVisionTransformer`.

Mirrors the interface of the reference's mmseg/models/backbones/mix_transformer.py (ctor kwargs :189-208, :538-551;
`forward(x[B,3,H,W]) -> list of 4 NCHW maps` :397-440; parameter names = its state_dict keys, so mit_b5.pth and CMDA
checkpoints load with load_state_dict).  Internally everything stays NLC/NHWC in the compute dtype; the NCHW maps
returned by `forward` are permuted *views* of the NHWC buffers (the reference's `.permute().contiguous()` copies at
:406,:414,:422,:430 do not exist here).
"""
import math

import torch
import torch.nn as nn

from . import nn as K
from . import ops
from . import runtime as rt
from .registry import BACKBONES


class DWConv(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, **_):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.dwconv = DWConv(hidden_features)
        self.fc2 = nn.Linear(hidden_features, out_features)

    def fwd(self, x, B, H, W):
        M, Cin = B * H * W, self.fc1.weight.shape[1]
        hidden = self.fc1.weight.shape[0]
        h = K.linear_fwd(x, self.fc1.weight, self.fc1.bias, M, Cin)
        dw = self.dwconv.dwconv
        act = ops.dwconv_fwd(h, rt.wdw(dw.weight), dw.bias, B, H, W, hidden, 1, 'gelu')
        y = K.linear_fwd(act, self.fc2.weight, self.fc2.bias, M, hidden)
        return y, (x, h, act)

    def bwd(self, saved, dy, B, H, W):
        x, h, act = saved
        return K.mlp_bwd(dy, self, x, h, act, B, H, W, self.fc1.weight.shape[1])


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, sr_ratio=1, **_):
        super().__init__()
        assert dim % num_heads == 0
        self.dim, self.num_heads, self.sr_ratio = dim, num_heads, sr_ratio
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        if sr_ratio > 1:
            self.sr = nn.Conv2d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)
            self.norm = nn.LayerNorm(dim)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=None, norm_layer=None, sr_ratio=1):
        super().__init__()
        assert qk_scale is None and drop == 0. and attn_drop == 0., 'only the configuration CMDA uses is implemented'
        self.dim, self.num_heads, self.sr_ratio, self.drop_path_rate = dim, num_heads, sr_ratio, float(drop_path)
        self.eps = getattr(norm_layer, 'keywords', None) and norm_layer.keywords.get('eps', 1e-5) or 1e-5
        self.norm1 = nn.LayerNorm(dim, eps=self.eps)
        self.attn = Attention(dim, num_heads, qkv_bias, sr_ratio)
        self.norm2 = nn.LayerNorm(dim, eps=self.eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def _dp(self, B, device):
        if not self.training or self.drop_path_rate == 0. or not getattr(self, 'stochastic', True):
            return None
        pool = getattr(self, '_dp_pool', None)
        if pool is not None and pool[0].shape[1] == B:  # rows drawn once per backbone pass (3 launches instead of 4 per call)
            masks, idx = pool
            pool[1] = idx + 1
            return masks[idx]
        keep = 1.0 - self.drop_path_rate
        return (keep + torch.rand(B, device=device)).floor_().div_(keep)

    def fwd(self, x, B, H, W, save=True):
        dp1, dp2 = self._dp(B, x.device), self._dp(B, x.device)
        return K.block_fwd(x, self, B, H, W, self.dim, self.num_heads, self.sr_ratio, eps=self.eps, dp1=dp1, dp2=dp2,
                           save=save)

    def bwd(self, saved, dy, B, H, W, dy_scaled=None, next_scale=None):
        with rt.lane_batch('wgrad'):   # the block's weight gradients: one entry of the side lane after its dgrad chain (if enabled)
            return K.block_bwd(dy, self, saved, B, H, W, self.dim, self.num_heads, self.sr_ratio, eps=self.eps, dy_scaled=dy_scaled,
                               next_scale=next_scale)

    def forward(self, x, H, W):
        """Reference signature: x [B,N,C] -> [B,N,C] (autograd-enabled bridge over fwd/bwd)."""
        return _BlockFn.apply(self, H, W, x, rt.anchor(x.device))


class _BlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, blk, H, W, x, anchor):
        B, N, C = x.shape
        xc = ops.cast(x.contiguous().view(B * N, C), rt.compute_dtype())
        y, saved = blk.fwd(xc, B, H, W)
        ctx.blk, ctx.saved, ctx.geo, ctx.in_dtype = blk, saved, (B, H, W, N, C), x.dtype
        return ops.cast(y, x.dtype).view(B, N, C)

    @staticmethod
    def backward(ctx, dy):
        B, H, W, N, C = ctx.geo
        dyc = ops.cast(dy.contiguous().view(B * N, C), rt.compute_dtype())
        dx = ctx.blk.bwd(ctx.saved, dyc, B, H, W)
        return None, None, None, ops.cast(dx, ctx.in_dtype).view(B, N, C), None


class OverlapPatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=7, stride=4, in_chans=3, embed_dim=768):
        super().__init__()
        self.patch_size, self.stride = patch_size, stride
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=stride, padding=patch_size // 2)
        self.norm = nn.LayerNorm(embed_dim)  # eps 1e-5 (torch default), as in the reference

    def fwd(self, x, B, H, W):
        """x [B*H*W, Cin] NHWC rows -- or Cin padded with zero channels (runtime.conv_channel_pad: the 3-channel input)"""
        cp = x.shape[1] if x.shape[1] > self.proj.weight.shape[1] else 0
        y, OH, OW = K.conv_fwd(x, self.proj.weight, self.proj.bias, B, H, W, self.stride, self.patch_size // 2, ci_pad=cp)
        # the norm's output opens the stage's residual stream: fp32 storage in the bf16 mode (runtime.residual_fp32)
        yn, m, r = ops.layernorm_fwd(y, self.norm.weight, self.norm.bias, 1e-5, out_dtype=rt.stream_dtype())
        return yn, OH, OW, (x, y, m, r, H, W)

    def bwd(self, saved, dyn, B, need_dx=True):
        x, y, m, r, H, W = saved
        dy = ops.layernorm_bwd(dyn, y, self.norm.weight, m, r, rt.grad(self.norm.weight), rt.grad(self.norm.bias))
        cp = x.shape[1] if x.shape[1] > self.proj.weight.shape[1] else 0
        return K.conv_bwd(dy, x, self.proj.weight, self.proj.bias, B, H, W, self.stride, self.patch_size // 2,
                          need_dx=need_dx, ci_pad=cp)


def draw_drop_path(owner, blocks, B, device):
    """Stochastic depth (mix_transformer.py:147-152, DropPath per residual branch) for a run of blocks that is about to execute: all
    of their per-sample keep masks come from ONE torch.rand call (4 launches per pass instead of 4 per mask); block i takes rows
    2i (attention branch) and 2i+1 (MLP branch).  `owner` caches the per-row keep probabilities."""
    live = [blk for blk in blocks if blk.training and blk.drop_path_rate > 0. and getattr(blk, 'stochastic', True)]
    if not live:
        return
    rates = tuple(blk.drop_path_rate for blk in live)
    cached = getattr(owner, '_dp_keep', None)
    if cached is None or cached[0] != (rates, str(device)):
        keep = torch.tensor([1.0 - r for r in rates for _ in range(2)], dtype=torch.float32).to(device).unsqueeze(1)
        owner._dp_keep = cached = ((rates, str(device)), keep)
    keep = cached[1]
    masks = (keep + torch.rand(2 * len(live), B, device=device)).floor_().div_(keep)
    rt.tap(('drop_path', getattr(owner, '_tap_name', type(owner).__name__)), masks)
    for i, blk in enumerate(live):
        blk._dp_pool = [masks, 2 * i]


@BACKBONES.register_module()
class MixVisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dims=[64, 128, 256, 512],
                 num_heads=[1, 2, 4, 8], mlp_ratios=[4, 4, 4, 4], qkv_bias=False, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0.1, norm_layer=None, depths=[3, 4, 6, 3], sr_ratios=[8, 4, 2, 1],
                 style=None, pretrained=None, init_cfg=None, freeze_patch_embed=False):
        super().__init__()
        assert not (init_cfg and pretrained), 'init_cfg and pretrained cannot be setting at the same time'
        if not (isinstance(pretrained, str) or pretrained is None):
            raise TypeError('pretrained must be a str or None')
        self.depths, self.pretrained, self.init_cfg = list(depths), pretrained, init_cfg
        self.embed_dims, self.num_heads, self.sr_ratios = list(embed_dims), list(num_heads), list(sr_ratios)
        self.eps = getattr(norm_layer, 'keywords', None) and norm_layer.keywords.get('eps', 1e-5) or 1e-5
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths), device='cpu')]
        cur = 0
        for s in range(4):
            cin = in_chans if s == 0 else embed_dims[s - 1]
            setattr(self, f'patch_embed{s + 1}', OverlapPatchEmbed(img_size // (1 if s == 0 else 2 ** (s + 1)),
                                                                    7 if s == 0 else 3, 4 if s == 0 else 2, cin,
                                                                    embed_dims[s]))
            blocks = nn.ModuleList([
                Block(embed_dims[s], num_heads[s], mlp_ratios[s], qkv_bias, qk_scale, drop_rate, attn_drop_rate,
                      dpr[cur + i], norm_layer=norm_layer, sr_ratio=sr_ratios[s]) for i in range(depths[s])])
            for b in blocks:
                b.eps = self.eps
                b.norm1.eps = b.norm2.eps = self.eps
            setattr(self, f'block{s + 1}', blocks)
            setattr(self, f'norm{s + 1}', nn.LayerNorm(embed_dims[s], eps=self.eps))
            cur += depths[s]
        if freeze_patch_embed:
            self.patch_embed1.requires_grad = False

    # -- weights ------------------------------------------------------------------------------------------
    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)
        elif isinstance(m, nn.Conv2d):
            fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
            m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
            if m.bias is not None:
                m.bias.data.zero_()

    def init_weights(self):
        if self.pretrained is None:
            for m in self.modules():
                self._init_weights(m)
        elif isinstance(self.pretrained, str):
            from .checkpoint import load_checkpoint  # mix_transformer.py:343-357: _load_checkpoint + non-strict load
            load_checkpoint(self, self.pretrained, map_location='cpu', strict=False)
        rt.invalidate()

    def reset_drop_path(self, drop_path_rate):
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(self.depths), device='cpu')]
        cur = 0
        for s in range(4):
            for i, blk in enumerate(getattr(self, f'block{s + 1}')):
                blk.drop_path_rate = dpr[cur + i]
            cur += self.depths[s]

    # -- hand-scheduled passes ------------------------------------------------------------------------------
    def _draw_drop_path(self, B, device):
        draw_drop_path(self, [blk for s in range(1, 5) for blk in getattr(self, f'block{s}')], B, device)

    def feature_shapes(self, H, W):
        """(H_s, W_s) of the four stage outputs for an H x W input (patch embeds: k7 s4 p3, then k3 s2 p1)"""
        out = []
        for s in range(4):
            k, st = (7, 4) if s == 0 else (3, 2)
            H, W = K.conv_out_size(H, W, k, st, k // 2)
            out.append((H, W))
        return out

    @ops.sited('mit')
    def fwd(self, img, save=True, out_feats=None):
        """img: NCHW fp32 [B,3,H,W] (the reference's input layout), or a LIST of such tensors run as ONE batch (the event
        encoder sees the events and the ISR of a sample with the same weights, encoder_decoder.py:703-712: one pass over 2B
        samples instead of two over B).  out_feats: optional list of 4 pre-allocated [Btot*N_s, C_s] tensors the stage outputs
        are written into (slices of the decode head's joint feature buffers).
        Returns ([(feat [Btot*N,C], H, W)] * 4, saved)."""
        imgs = list(img) if isinstance(img, (list, tuple)) else [img]
        _, Cin, H, W = imgs[0].shape
        B = sum(t.shape[0] for t in imgs)
        cp = rt.conv_channel_pad(Cin)
        x = torch.empty(B * H * W, cp, dtype=rt.compute_dtype(), device=imgs[0].device)
        row = 0
        for t in imgs:
            b = t.shape[0]
            if cp != Cin:
                ops.nchw_to_nhwc_pad(t.contiguous(), x[row:row + b * H * W], b, Cin, H * W, cp)
            else:
                ops.permute4(t.contiguous(), x[row:row + b * H * W], (b, Cin, H, W), (0, 2, 3, 1))
            row += b * H * W
        feats, saved = [], []
        self._draw_drop_path(B, x.device)
        for s in range(1, 5):
            pe = getattr(self, f'patch_embed{s}')
            x, H, W, sv_pe = pe.fwd(x, B, H, W)
            sv_blocks = []
            for blk in getattr(self, f'block{s}'):
                x, sv = blk.fwd(x, B, H, W, save=save)
                blk._dp_pool = None
                sv_blocks.append(sv)
            nrm = getattr(self, f'norm{s}')
            xin = x
            x, m, r = ops.layernorm_fwd(xin, nrm.weight, nrm.bias, self.eps, out=out_feats[s - 1] if out_feats is not None else None,
                                        out_dtype=rt.compute_dtype())   # stage output: back to the compute dtype
            feats.append((x, H, W))
            saved.append((sv_pe, sv_blocks, (xin, m, r), H, W))
        return feats, (saved, B) if save else None

    @ops.sited('mit')
    def bwd(self, saved_all, dfeats):
        """dfeats: list of 4 gradients [B*N_s, C_s] (compute dtype; None = zero).  Accumulates parameter gradients."""
        saved, B = saved_all
        dnext = None  # gradient flowing into stage s's output from stage s+1's patch embed
        for s in range(4, 0, -1):
            sv_pe, sv_blocks, (xin, m, r), H, W = saved[s - 1]
            d = dfeats[s - 1]
            if d is None:
                d = dnext
            elif dnext is not None:
                d = ops.axpby(d, dnext, 1.0, 1.0)
            if d is None:
                dnext = None
                continue
            nrm = getattr(self, f'norm{s}')
            blocks = getattr(self, f'block{s}')
            # every LayerNorm backward also writes its result scaled by the DropPath factor of the block that consumes it next
            # (saved[-1] = that block's MLP-branch factor), so no separate scaling kernel runs
            scales = [sv[-1] for sv in sv_blocks]
            N = H * W
            dx = ops.layernorm_bwd(d, xin, nrm.weight, m, r, rt.grad(nrm.weight), rt.grad(nrm.bias), out_scale=scales[-1],
                                   rows_per_scale=N)
            dxs = None
            if scales[-1] is not None:
                dx, dxs = dx
            for i in range(len(blocks) - 1, -1, -1):
                nxt = scales[i - 1] if i > 0 else None
                dx = blocks[i].bwd(sv_blocks[i], dx, B, H, W, dy_scaled=dxs, next_scale=nxt)
                dxs = None
                if nxt is not None:
                    dx, dxs = dx
            dnext = getattr(self, f'patch_embed{s}').bwd(sv_pe, dx, B, need_dx=(s > 1))
            # the stage's queued weight gradients: one grouped launch per (tile, operand mode).  Lane 'wq' (uda.DACS, single GPU): they
            # leave this lane's dependent dgrad chain and run on a side queue underneath the next stage's chain -- the queues the teacher
            # used in the first part of the iteration are idle by now (runtime.LANE_ALIAS).  With a gradient exchange armed the stage's
            # gradients must be final here: flushed in place.
            if rt.grad_ready_hook is None:
                src_lane = ops.LN_LANE
                with rt.lane('wq', *ops.gemm_deferred_tensors()):
                    ops.gemm_flush_deferred(from_lane=src_lane)
            else:
                ops.gemm_flush_deferred()
            rt.notify_grads_ready(f'backbone.stage{s}', self)
        rt.join_lanes('wq')
        return None

    def forward(self, x):
        return list(_BackboneFn.apply(self, x, rt.anchor(x.device), torch.is_grad_enabled()))


class _BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, img, anchor, need_grad):
        feats, saved = net.fwd(img.float(), save=need_grad)
        ctx.net, ctx.saved = net, saved
        B = img.shape[0]
        ctx.shapes = [(B, H, W, f.shape[1]) for f, H, W in feats]
        return tuple(ops.cast(f, torch.float32).view(B, H, W, -1).permute(0, 3, 1, 2) for f, H, W in feats)

    @staticmethod
    def backward(ctx, *grads):
        dfeats = []
        for g, (B, H, W, C) in zip(grads, ctx.shapes):
            if g is None:
                dfeats.append(None)
                continue
            t = torch.empty(B * H * W, C, dtype=rt.compute_dtype(), device=g.device)
            ops.permute4(g.contiguous(), t, (B, C, H, W), (0, 2, 3, 1))
            dfeats.append(t)
        ctx.net.bwd(ctx.saved, dfeats)
        return None, None, None, None


def _variant(name, embed_dims, depths):
    def __init__(self, **kwargs):
        from functools import partial
        MixVisionTransformer.__init__(self, patch_size=4, embed_dims=embed_dims, num_heads=[1, 2, 5, 8],
                                      mlp_ratios=[4, 4, 4, 4], qkv_bias=True,
                                      norm_layer=partial(nn.LayerNorm, eps=1e-6), depths=depths,
                                      sr_ratios=[8, 4, 2, 1], **kwargs)
    cls = type(name, (MixVisionTransformer,), {'__init__': __init__, '__doc__': f'{name} (mix_transformer.py:460-551)'})
    return BACKBONES.register_module()(cls)


mit_b0 = _variant('mit_b0', [32, 64, 160, 256], [2, 2, 2, 2])
mit_b1 = _variant('mit_b1', [64, 128, 320, 512], [2, 2, 2, 2])
mit_b2 = _variant('mit_b2', [64, 128, 320, 512], [3, 4, 6, 3])
mit_b3 = _variant('mit_b3', [64, 128, 320, 512], [3, 4, 18, 3])
mit_b4 = _variant('mit_b4', [64, 128, 320, 512], [3, 8, 27, 3])
mit_b5 = _variant('mit_b5', [64, 128, 320, 512], [3, 6, 40, 3])
