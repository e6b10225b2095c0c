"""DAFormer decode heads on the HIP kernels -- registry keys `DAFormerHead`, `DAFormerHeadFusion`, `CrossEntropyLoss`.

Mirrors mmseg/models/decode_heads/daformer_head.py (DAFormerHead :136-197, DAFormerHeadFusion :200-322),
decode_head.py (BaseDecodeHead(.Fusion): ctor :49-97 / :275-343, cls_seg* :563-586, losses :588-606,
forward_train :423-534), aspp_head.py / sep_aspp_head.py (ASPP modules) and losses/cross_entropy_loss.py:140-200.
Parameter names equal the reference's state_dict keys (mmcv ConvModule: `.conv` / `.bn`).

Data layout: every feature map is NHWC flattened to [B*H*W, C]; the four embeds are written straight into their
channel slice of one [B*H*W, 1024] buffer (no torch.cat), the four ASPP branches likewise into the bottleneck's input;
logits stay fp32 NHWC and the 19 x H x W up-sampled tensor of losses() is never materialised (ce_loss.hip).
"""
import torch
import torch.nn as nn

from . import nn as K
from . import ops
from . import runtime as rt
from .registry import HEADS, LOSSES, build_loss


# ---------------------------------------------------------------------------------------------- containers
class ConvModule(nn.Module):
    """Parameter container with mmcv.cnn.ConvModule's naming: conv (no bias when a norm follows) + bn (+ReLU)."""

    def __init__(self, cin, cout, k, padding=0, dilation=1, groups=1, norm=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=padding, dilation=dilation, groups=groups, bias=not norm)
        self.bn = nn.BatchNorm2d(cout) if norm else None
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode='fan_out', nonlinearity='relu')
        if self.conv.bias is not None:
            nn.init.zeros_(self.conv.bias)
        self.k, self.padding, self.dilation, self.groups = k, padding, dilation, groups


class DepthwiseSeparableConvModule(nn.Module):
    def __init__(self, cin, cout, k, padding, dilation):
        super().__init__()
        self.depthwise_conv = ConvModule(cin, cin, k, padding=padding, dilation=dilation, groups=cin)
        self.pointwise_conv = ConvModule(cin, cout, 1)


class MLP(nn.Module):
    def __init__(self, input_dim=2048, embed_dim=768):
        super().__init__()
        self.proj = nn.Linear(input_dim, embed_dim)


def _bn_stats_ws(bn, dev, M, C, groups):
    """the workspace in which the convolution in front of `bn` leaves the batch statistics (its GEMM epilogue, ops.gemm colstats), or
    None: evaluation mode, groups that are not whole 256-row tiles, fused statistics switched off"""
    if not bn.training or M % groups or not ops.colstats_ok(M // groups, C):
        return None
    return ops.bn_stats_ws(dev, groups, C)


def _bn_fwd(bn, x, y, M, C, relu, ldy=None, coff=0, groups=1, order=None, stats_ws=None):
    """train: batch stats + running-stat update; eval: running stats.  Returns what the backward needs.  M = ALL rows of x;
    groups > 1: x is `groups` consecutive blocks of M/groups rows, each with its own batch statistics (one decoder pass over
    several feature sets, running statistics updated in `order`).  stats_ws: `_bn_stats_ws` filled by the producer of x."""
    if bn.training:
        mean, rstd = ops.bn_train_fwd(x, bn.weight, bn.bias, y, bn.running_mean, bn.running_var, M // groups, C, bn.eps,
                                      bn.momentum, relu, ldy, coff, groups, order, stats_ws=stats_ws)
    else:
        mean, rstd = bn.running_mean, torch.rsqrt(bn.running_var + bn.eps)
        ops.bn_apply(x, mean, rstd, bn.weight, bn.bias, y, M, C, relu, ldy, coff)
        mean, rstd = mean.expand(groups, C).contiguous(), rstd.expand(groups, C).contiguous()
    return mean, rstd


def _bn_bwd(bn, dy, x, stats, M, C, relu, lddy=None, coff=0, groups=1):
    mean, rstd = stats
    return ops.bn_train_bwd(dy, x, mean, rstd, bn.weight, bn.bias, rt.grad(bn.weight), rt.grad(bn.bias), M // groups, C, relu,
                            lddy, coff, groups)


class ASPPWrapper(nn.Module):
    """daformer_head.py:15-79 with pool=False, no context layer (configs/_base_/models/daformer_sepaspp_mitb5.py)."""

    def __init__(self, in_channels, channels, sep, dilations, pool=False, norm_cfg=None, act_cfg=None,
                 align_corners=False, context_cfg=None):
        super().__init__()
        assert not pool and context_cfg is None, 'only the configuration CMDA uses is implemented'
        assert sep, 'CMDA uses the depthwise-separable ASPP'
        self.dilations = tuple(dilations)
        self.in_channels, self.channels = in_channels, channels
        mods = []
        for d in self.dilations:
            mods.append(ConvModule(in_channels, channels, 1) if d == 1 else
                        DepthwiseSeparableConvModule(in_channels, channels, 3, padding=d, dilation=d))
        self.aspp_modules = nn.ModuleList(mods)
        self.bottleneck = ConvModule(len(self.dilations) * channels, channels, 3, padding=1)

    def fwd(self, x, B, H, W, groups=1, order=None):
        """x [M, Cin] -> feat [M, channels]; B = all images in x; groups / order: see _bn_fwd"""
        M, Cin, Ch = B * H * W, self.in_channels, self.channels
        nb = len(self.dilations)
        cat = torch.empty(M, nb * Ch, dtype=rt.compute_dtype(), device=x.device)
        saved = []
        g = dict(groups=groups, order=order)
        for j, (d, m) in enumerate(zip(self.dilations, self.aspp_modules)):
            if d == 1:
                ws = _bn_stats_ws(m.bn, x.device, M, Ch, groups)
                z = K.linear_fwd(x, m.conv.weight, None, M, Cin, colstats=None if ws is None else (ws, M // groups))
                st = _bn_fwd(m.bn, z, cat, M, Ch, True, nb * Ch, j * Ch, stats_ws=ws, **g)
                saved.append((z, st))
            else:
                dwm, pwm = m.depthwise_conv, m.pointwise_conv
                ws = _bn_stats_ws(dwm.bn, x.device, M, Cin, groups) if (d >= 2 and B % groups == 0) else None
                u = ops.dwconv_fwd(x, rt.wdw(dwm.conv.weight), None, B, H, W, Cin, d, None, colstats=None if ws is None else (ws, B // groups))
                ub = torch.empty_like(u)
                st_u = _bn_fwd(dwm.bn, u, ub, M, Cin, True, stats_ws=ws, **g)
                ws = _bn_stats_ws(pwm.bn, x.device, M, Ch, groups)
                z = K.linear_fwd(ub, pwm.conv.weight, None, M, Cin, colstats=None if ws is None else (ws, M // groups))
                st_z = _bn_fwd(pwm.bn, z, cat, M, Ch, True, nb * Ch, j * Ch, stats_ws=ws, **g)
                saved.append((u, st_u, ub, z, st_z))
        ws = _bn_stats_ws(self.bottleneck.bn, x.device, M, Ch, groups)
        zb, _, _ = K.conv_fwd(cat, self.bottleneck.conv.weight, None, B, H, W, 1, 1, colstats=None if ws is None else (ws, M // groups))
        feat = torch.empty_like(zb)
        st_b = _bn_fwd(self.bottleneck.bn, zb, feat, M, Ch, True, stats_ws=ws, **g)
        return feat, (x, cat, saved, zb, st_b, groups)

    def bwd(self, sv, dfeat, B, H, W):
        x, cat, saved, zb, st_b, groups = sv
        M, Cin, Ch = B * H * W, self.in_channels, self.channels
        nb = len(self.dilations)
        dzb = _bn_bwd(self.bottleneck.bn, dfeat, zb, st_b, M, Ch, True, groups=groups)
        dcat = K.conv_bwd(dzb, cat, self.bottleneck.conv.weight, None, B, H, W, 1, 1)
        dx = torch.empty(M, Cin, dtype=rt.compute_dtype(), device=x.device)
        first = True
        for j, (d, m) in enumerate(zip(self.dilations, self.aspp_modules)):
            if d == 1:
                z, st = saved[j]
                dz = _bn_bwd(m.bn, dcat, z, st, M, Ch, True, nb * Ch, j * Ch, groups=groups)
                K.linear_bwd(dz, x, m.conv.weight, None, M, Cin, dx_out=dx, dx_beta=0.0 if first else 1.0)
            else:
                u, st_u, ub, z, st_z = saved[j]
                dwm, pwm = m.depthwise_conv, m.pointwise_conv
                dz = _bn_bwd(pwm.bn, dcat, z, st_z, M, Ch, True, nb * Ch, j * Ch, groups=groups)
                dub = K.linear_bwd(dz, ub, pwm.conv.weight, None, M, Cin)
                du = _bn_bwd(dwm.bn, dub, u, st_u, M, Cin, True, groups=groups)
                # (off the critical path: with a tail queue open -- segmentors.train_bwd -- it runs in the tail of the backward pass)
                ops.tail_defer(lambda du=du, gw=rt.grad(dwm.conv.weight).view(Cin, 9), d=d:
                               ops.dwconv_bwd_weight(du, x, gw, None, B, H, W, Cin, d))
                ops.dwconv_bwd_data(du, rt.wdw(dwm.conv.weight), B, H, W, Cin, d, out=dx, accumulate=not first)
            first = False
        return dx


def build_layer(in_channels, out_channels, type, **kwargs):
    """daformer_head.py:82-116 (the layer types CMDA's configs use)."""
    if type == 'id':
        return nn.Identity()
    if type == 'mlp':
        return MLP(input_dim=in_channels, embed_dim=out_channels)
    if type == 'aspp':
        return ASPPWrapper(in_channels=in_channels, channels=out_channels, **kwargs)
    raise NotImplementedError(type)


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    """losses/cross_entropy_loss.py:140-200, softmax path (use_sigmoid=False, use_mask=False, no class weights)."""

    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None, loss_weight=1.0):
        super().__init__()
        assert not use_sigmoid and not use_mask, 'only the softmax cross-entropy of configs/fusion/* is implemented'
        assert reduction == 'mean'
        self.use_sigmoid, self.use_mask, self.reduction = use_sigmoid, use_mask, reduction
        self.loss_weight, self.class_weight = loss_weight, class_weight


def ce_losses_fwd(logits, seg_label, seg_weight, ignore_index, loss_weight):
    """losses(): logits fp32 NHWC [B,h,w,nc]; label [B,1,H,W] int64.  Returns (loss, acc%, saved)."""
    B, _, H, W = seg_label.shape
    label = seg_label.view(B, H, W)
    wgt = seg_weight.float().contiguous() if seg_weight is not None else None
    acc, lse = ops.ce_upsample_fwd(logits, label, wgt, H, W, ignore_index)
    n = float(B * H * W)
    loss = acc[0] * (loss_weight / n)
    accuracy = acc[1:2] * (100.0 / n)
    return loss, accuracy, (logits, label, wgt, lse, H, W, n)


def ce_losses_bwd(saved, gscale, mul, ignore_index, loss_weight, out=None):
    """d(loss)/d(logits) * gscale (fp32 device scalar or None) * mul"""
    logits, label, wgt, lse, H, W, n = saved
    return ops.ce_upsample_bwd(logits, label, wgt, lse, gscale, mul * loss_weight / n, H, W, ignore_index, out=out)


# ---------------------------------------------------------------------------------------------- heads
class _HeadBase(nn.Module):
    def __init__(self, in_channels, channels, *, num_classes, dropout_ratio=0.1, conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type='ReLU'), in_index=-1, input_transform=None,
                 loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0), decoder_params=None,
                 ignore_index=255, sampler=None, align_corners=False, init_cfg=None):
        super().__init__()
        assert input_transform == 'multiple_select' and sampler is None
        assert isinstance(in_channels, (list, tuple)) and isinstance(in_index, (list, tuple))
        assert len(in_channels) == len(in_index)
        assert not align_corners
        self.in_channels, self.in_index, self.channels = list(in_channels), list(in_index), channels
        self.num_classes, self.dropout_ratio, self.ignore_index = num_classes, dropout_ratio, ignore_index
        self.norm_cfg, self.act_cfg, self.align_corners = norm_cfg, act_cfg, align_corners
        self.input_transform = input_transform
        self.loss_decode = build_loss(loss_decode)
        self.conv_seg = nn.Conv2d(channels, num_classes, kernel_size=1)
        nn.init.normal_(self.conv_seg.weight, std=0.01)
        nn.init.zeros_(self.conv_seg.bias)
        dp = decoder_params
        embed_dims = dp['embed_dims']
        self.embed_dims = [embed_dims] * len(self.in_index) if isinstance(embed_dims, int) else list(embed_dims)
        embed_cfg, neck_cfg = dp['embed_cfg'], dp['embed_neck_cfg']
        self._embed_cfg = dict(embed_cfg)
        self._neck_cfg = dict(embed_cfg if neck_cfg == 'same_as_embed_cfg' else neck_cfg)
        self._fusion_cfg = dict(dp['fusion_cfg'])
        for cfg in (self._embed_cfg, self._neck_cfg, self._fusion_cfg):
            if 'aspp' in cfg['type']:
                cfg['align_corners'] = self.align_corners

    def init_weights(self):
        pass  # constructors already apply the reference's initialisers

    def _make_branch(self):
        embeds = {}
        for i, cin, e in zip(self.in_index, self.in_channels, self.embed_dims):
            cfg = self._neck_cfg if i == self.in_index[-1] else self._embed_cfg
            embeds[str(i)] = build_layer(cin, e, **cfg)
        return nn.ModuleDict(embeds), build_layer(sum(self.embed_dims), self.channels, **self._fusion_cfg)

    # one decoder branch: embeds -> resize+concat -> fuse layer
    def _branch_fwd(self, embeds, fuse, feats, B, groups=1, order=None):
        """B = all images in `feats` (groups * per-branch batch when several feature sets share the decoder weights)"""
        f0, H, W = feats[self.in_index[0]]
        M = B * H * W
        tot = sum(self.embed_dims)
        cat = torch.empty(M, tot, dtype=rt.compute_dtype(), device=f0.device)
        off = 0
        for i, e in zip(self.in_index, self.embed_dims):
            f, h, w = feats[i]
            lin = embeds[str(i)].proj
            if (h, w) == (H, W):
                K.linear_fwd(f, lin.weight, lin.bias, B * h * w, f.shape[1], out=cat, ldc=tot, c_offset=off)
            else:
                emb = K.linear_fwd(f, lin.weight, lin.bias, B * h * w, f.shape[1])
                ops.bilinear_fwd(emb, cat, B, h, w, H, W, e, tot, off)
            off += e
        feat, sv = fuse.fwd(cat, B, H, W, groups, order) if groups > 1 else fuse.fwd(cat, B, H, W)
        return feat, (sv, feats, H, W)

    def _branch_bwd(self, embeds, fuse, saved, dfeat, B):
        sv, feats, H, W = saved
        tot = sum(self.embed_dims)
        dcat = fuse.bwd(sv, dfeat, B, H, W)
        dfs = {}
        off = 0
        for i, e in zip(self.in_index, self.embed_dims):
            f, h, w = feats[i]
            lin = embeds[str(i)].proj
            if (h, w) == (H, W):
                dfs[i] = K.linear_bwd(dcat, f, lin.weight, lin.bias, B * h * w, f.shape[1], dy_ld=tot, dy_off=off)
            else:
                demb = torch.empty(B * h * w, e, dtype=rt.compute_dtype(), device=dcat.device)
                ops.bilinear_bwd(dcat, demb, B, h, w, H, W, e, tot, off)
                dfs[i] = K.linear_bwd(demb, f, lin.weight, lin.bias, B * h * w, f.shape[1])
            off += e
        return dfs

    # classifier: (Dropout2d) -> 1x1 conv, logits fp32 NHWC
    def _cls_fwd(self, feat, B, H, W, with_dropout=True, drop_B=None):
        """drop_B: only the first drop_B samples get Dropout2d (the image branch of a grouped pass; decode_head.py:570-586:
        cls_seg has the dropout, cls_seg_events / _fusion do not)"""
        M, Ch = B * H * W, self.channels
        mask = None
        if with_dropout and self.training and self.dropout_ratio > 0 and getattr(self, 'stochastic', True):
            keep = 1.0 - self.dropout_ratio
            forced = getattr(self, 'inject_dropout_mask', None)   # tests: the oracle's Dropout2d mask [drop_B or B, Ch] of 0/1
            nb = B if drop_B is None else drop_B
            m = (torch.rand(nb, Ch, device=feat.device) < keep).float() if forced is None else forced.to(feat.device).float()
            rt.tap(('dropout2d', getattr(self, '_tap_name', type(self).__name__)), m)
            m = m / keep
            if nb < B:
                mask = torch.ones(B, Ch, dtype=torch.float32, device=feat.device)
                mask[:nb].copy_(m)
            else:
                mask = m
            featd = ops.sample_scale(feat, mask, B, Ch, per_channel=True)
        else:
            featd = feat
        logits = K.linear_fwd(featd, self.conv_seg.weight, self.conv_seg.bias, M, Ch,
                              out_dtype=torch.float32)
        return logits.view(B, H, W, self.num_classes), (featd, mask)

    def _cls_bwd(self, saved, dlogits, B, H, W):
        featd, mask = saved
        M, Ch = B * H * W, self.channels
        dl = dlogits.view(M, self.num_classes)
        nc = self.num_classes
        if rt.compute_dtype() == torch.float32:
            dfeat = K.linear_bwd(dl, featd, self.conv_seg.weight, self.conv_seg.bias, M, Ch)
        elif nc % 8 == 0:
            dfeat = K.linear_bwd(ops.cast(dl, rt.compute_dtype()), featd, self.conv_seg.weight, self.conv_seg.bias, M, Ch)
        else:
            # 19 classes: rows of 38 bytes would send both GEMMs to the register-staged kernel (141 + 180 us at 262144 pixels); the
            # cast that had to run anyway pads the rows to 32 columns of zeros instead, the classifier weight keeps its 19 rows (the
            # operand views read zero past them)
            ncp = (nc + 31) // 32 * 32
            dlp = ops.cast_pad_cols(dl, ncp, rt.compute_dtype())
            w, b = self.conv_seg.weight, self.conv_seg.bias
            dlv = ops.plain_view(dlp, M, ncp)
            ops.gemm(dlv, ops.plain_view(featd, M, Ch), rt.grad(w).view(nc, Ch), nc, Ch, M, a_kstrided=True, b_kstrided=True, dtype=rt.tag(),
                     atomic=True, splits=0, colsum=rt.grad(b), defer=True, keep=(dlp, featd))
            dfeat = torch.empty(M, Ch, dtype=rt.compute_dtype(), device=dl.device)
            ops.gemm(dlv, ops.plain_view(rt.w(w).view(nc, Ch), nc, Ch), dfeat, M, Ch, ncp, b_kstrided=True, dtype=rt.tag())
        if mask is not None:
            dfeat = ops.sample_scale(dfeat, mask, B, Ch, per_channel=True, out=dfeat)
        return dfeat

    @staticmethod
    def _to_feats(inputs):
        """NCHW list (reference interface) -> [(NLC tensor, H, W)] in the compute dtype."""
        out = []
        for x in inputs:
            B, C, H, W = x.shape
            t = torch.empty(B * H * W, C, dtype=rt.compute_dtype(), device=x.device)
            ops.permute4(x.contiguous(), t, (B, C, H, W), (0, 2, 3, 1))
            out.append((t, H, W))
        return out


@HEADS.register_module()
class DAFormerHead(_HeadBase):
    def __init__(self, **kwargs):
        super().__init__(input_transform='multiple_select', **kwargs)
        self.embed_layers, self.fuse_layer = self._make_branch()

    @ops.sited('head')
    def fwd(self, feats, B):
        feat, sv_b = self._branch_fwd(self.embed_layers, self.fuse_layer, feats, B)
        H, W = sv_b[2], sv_b[3]
        logits, sv_c = self._cls_fwd(feat, B, H, W)
        return logits, (sv_b, sv_c, H, W)

    @ops.sited('head')
    def bwd(self, saved, dlogits, B):
        sv_b, sv_c, H, W = saved
        dfeat = self._cls_bwd(sv_c, dlogits, B, H, W)
        return self._branch_bwd(self.embed_layers, self.fuse_layer, sv_b, dfeat, B)

    def fwd_train(self, feats, B, gt, seg_weight=None):
        logits, sv = self.fwd(feats, B)
        loss, acc, sv_l = ce_losses_fwd(logits, gt, seg_weight, self.ignore_index, self.loss_decode.loss_weight)
        return {'loss_seg': loss, 'acc_seg': acc}, logits, (sv, sv_l)

    def bwd_train(self, saved, B, gscale=None, mul=1.0):
        sv, sv_l = saved
        dlogits = ce_losses_bwd(sv_l, gscale, mul, self.ignore_index, self.loss_decode.loss_weight)
        return self.bwd(sv, dlogits, B)

    def forward(self, inputs):
        """Reference signature: list of NCHW maps -> logits [B,nc,h,w] (fp32, NCHW view). Inference-style (no grad)."""
        B = inputs[0].shape[0]
        logits, _ = self.fwd(self._to_feats([inputs[i] for i in range(len(inputs))]), B)
        return logits.permute(0, 3, 1, 2)


@HEADS.register_module()
class DAFormerHeadFusion(_HeadBase):
    """daformer_head.py:200-322 + BaseDecodeHeadFusion.forward_train decode_head.py:423-534 (the loss mix used by
    configs/fusion/*: not `cal_confidence`, not the `_split` train types)."""

    def __init__(self, **kwargs):
        dp = kwargs['decoder_params']
        assert 'train_type' in dp
        self.train_type = dp['train_type']
        assert self.train_type not in ('cs2dz_image+raw-isr_split', 'cs2dz_image+raw-isr_no-fusion'), \
            'split-classifier train types are outside the hot path (SURVEY.md section 2, row 12)'
        super().__init__(input_transform='multiple_select', **kwargs)
        self.split_cls = False
        self.share_decoder = bool(dp.get('share_decoder'))
        self.half_share_decoder = bool(dp.get('half_share_decoder'))
        assert not (self.share_decoder and self.half_share_decoder)
        self.embed_layers_image, self.fuse_layer_image = self._make_branch()
        self.embed_layers_events, self.fuse_layer_events = self._make_branch()
        self.embed_layers_fusion, self.fuse_layer_fusion = self._make_branch()
        if self.half_share_decoder:
            self.fuse_layer_events = self.fuse_layer_image
            self.fuse_layer_fusion = self.fuse_layer_image
        elif self.share_decoder:
            self.embed_layers_events = self.embed_layers_image
            self.fuse_layer_events = self.fuse_layer_image
            self.embed_layers_fusion = self.embed_layers_image
            self.fuse_layer_fusion = self.fuse_layer_image

    _BRANCHES = (('image_output', 'f_image', 'image', True), ('events_output', 'f_events', 'events', False),
                 ('fusion_output', 'f_fusion', 'fusion', False), ('img_self_res_output', 'f_img_self_res', 'events', False))

    def _layers(self, which):
        return getattr(self, f'embed_layers_{which}'), getattr(self, f'fuse_layer_{which}')

    @ops.sited('head')
    def fwd(self, inputs, B):
        """inputs: dict f_image / f_events / f_fusion / f_img_self_res -> list of (NLC tensor, H, W) or None."""
        out, saved = {}, {}
        for key, fkey, which, dropout in self._BRANCHES:
            feats = inputs.get(fkey)
            if feats is None:
                out[key] = None
                continue
            emb, fuse = self._layers(which)
            feat, sv_b = self._branch_fwd(emb, fuse, feats, B)
            H, W = sv_b[2], sv_b[3]
            logits, sv_c = self._cls_fwd(feat, B, H, W, with_dropout=dropout)
            out[key] = logits
            saved[key] = (sv_b, sv_c, H, W, which)
        return out, saved

    @ops.sited('head')
    def bwd(self, saved, dlogits, B):
        """dlogits: dict key -> gradient (or None).  Returns dict f_* -> {level: d feat}."""
        dfeats = {}
        for key, fkey, which, _ in self._BRANCHES:
            if key not in saved or dlogits.get(key) is None:
                continue
            sv_b, sv_c, H, W, which = saved[key]
            emb, fuse = self._layers(which)
            dfeat = self._cls_bwd(sv_c, dlogits[key], B, H, W)
            dfeats[fkey] = self._branch_bwd(emb, fuse, sv_b, dfeat, B)
        return dfeats

    # -- one pass of the shared decoder over ALL feature sets --------------------------------------------------------------
    # With share_decoder the four branches run the SAME weights (daformer_head.py:254-258), so their feature sets travel as one
    # batch of G*B images: rows [g*B*N, (g+1)*B*N) of each level's joint buffer belong to branch names[g].  GEMMs, depthwise
    # stencils and the 3x3 bottleneck see 4x larger grids (and a quarter of the launches); BatchNorm keeps per-branch batch
    # statistics (groups) and applies the running-statistic updates in the reference's branch order; Dropout2d touches the image
    # branch only.  Results equal the per-branch loop (tests/test_modules.py::test_head_fusion_joint_equals_per_branch).
    _KEY = {'image': 'image_output', 'fusion': 'fusion_output', 'events': 'events_output', 'isr': 'img_self_res_output'}
    _REF_ORDER = ('image', 'events', 'fusion', 'isr')   # order forward() runs the branches in (daformer_head.py:305-319)

    def joint_ok(self):
        return self.share_decoder and isinstance(self.fuse_layer_image, ASPPWrapper)

    @ops.sited('head')
    def fwd_joint(self, joint, names, B, passes=1):
        """joint: list of 4 (J_l [G*P*B*N_l, C_l], H_l, W_l); names: the G branch names in joint-buffer order, 'image' first.
        passes = P > 1: every branch block holds the B samples of P independent forward passes one after the other (the source
        and the mixed pass of one DACS iteration run the same weights, dacs.py:489-523 / :820-860): BatchNorm then keeps G*P
        groups of B samples and updates the running statistics pass by pass, branch by branch, as P separate calls would.
        Returns the logits of pass 0 as a dict (P == 1) or a list of P dicts."""
        G, P = len(names), passes
        assert names[0] == 'image' and self.joint_ok()
        order = sorted(range(G), key=lambda g: self._REF_ORDER.index(names[g]))
        order = [g * P + p for p in range(P) for g in order]
        emb, fuse = self._layers('image')
        feat, sv_b = self._branch_fwd(emb, fuse, joint, G * P * B, groups=G * P, order=order)
        H, W = sv_b[2], sv_b[3]
        logits, sv_c = self._cls_fwd(feat, G * P * B, H, W, with_dropout=True, drop_B=P * B)
        outs = []
        for p in range(P):
            out = {k: None for k in self._KEY.values()}
            for g, n in enumerate(names):
                out[self._KEY[n]] = logits[(g * P + p) * B:(g * P + p + 1) * B]
            outs.append(out)
        return (outs[0] if P == 1 else outs), (sv_b, sv_c, H, W, tuple(names), logits, P)

    @ops.sited('head')
    def bwd_joint(self, saved, dlogits_joint, B):
        """dlogits_joint fp32 [G*P*B,h,w,nc] -> {level: d joint features [G*P*B*N_l, C_l]}"""
        sv_b, sv_c, H, W, names, _, P = saved
        G = len(names)
        emb, fuse = self._layers('image')
        dfeat = self._cls_bwd(sv_c, dlogits_joint, G * P * B, H, W)
        return self._branch_bwd(emb, fuse, sv_b, dfeat, G * P * B)

    def _loss_mix(self, logits, gt, seg_weight, cfg):
        """BaseDecodeHeadFusion.forward_train's loss mix, decode_head.py:508-528"""
        lw = cfg['loss_weight']
        ii, lwt = self.ignore_index, self.loss_decode.loss_weight
        if seg_weight is None:
            seg_weight = torch.ones(gt.shape[0], gt.shape[2], gt.shape[3], dtype=torch.float32, device=gt.device)
        terms, sv_l = {}, {}
        for key in ('image_output', 'events_output', 'fusion_output', 'img_self_res_output'):
            if logits[key] is not None:
                loss, acc, sv = ce_losses_fwd(logits[key], gt, seg_weight, ii, lwt)
                terms[key] = (loss, acc)
                sv_l[key] = sv
        coef = {'image_output': lw['image']}
        if 'fusion_output' in terms:
            coef['fusion_output'] = lw['fusion']
        if 'img_self_res_output' in terms:
            coef['img_self_res_output'] = lw['img_self_res']
            coef['events_output'] = lw['events'] / 2
        else:
            coef['events_output'] = lw['events']
        # the reference accumulates fusion, image, then (isr, events): keep that order of fp32 additions
        order = [k for k in ('fusion_output', 'image_output', 'img_self_res_output', 'events_output') if k in coef]
        total = None
        for k in order:
            t = terms[k][0] * coef[k]
            total = t if total is None else total + t
        acc = terms['fusion_output'][1] if 'fusion_output' in terms else terms['image_output'][1]
        return {'loss_seg': total, 'acc_seg': acc}, sv_l, coef

    def fwd_train_joint(self, joint, names, B, gt, seg_weight=None, cfg=None, passes=1):
        """passes > 1: gt / seg_weight are lists (one per pass); returns lists of losses and logits.
        The G*P cross-entropy terms accumulate into ONE zeroed buffer and the loss mix of BaseDecodeHeadFusion.forward_train
        (decode_head.py:508-528) is a dot product per pass: 5 small launches instead of ~45."""
        logits, saved = self.fwd_joint(joint, names, B, passes)
        P, G = passes, len(names)
        gts = [gt] if P == 1 else list(gt)
        wts = [seg_weight] if P == 1 else list(seg_weight)
        lgs = [logits] if P == 1 else logits
        dev = gts[0].device
        ii, lwt = self.ignore_index, self.loss_decode.loss_weight
        H, W = gts[0].shape[2:]
        n = float(B * H * W)
        lw = cfg['loss_weight']
        coef = {'image_output': lw['image'], 'fusion_output': lw['fusion']}
        if 'isr' in names:
            coef['img_self_res_output'] = lw['img_self_res']
            coef['events_output'] = lw['events'] / 2
        else:
            coef['events_output'] = lw['events']
        acc = torch.zeros(G * P, 2, dtype=torch.float32, device=dev)
        sv_ls = []
        for p in range(P):
            label = gts[p].view(B, H, W)
            wgt = wts[p].float().contiguous() if wts[p] is not None else None   # None = weight 1 (decode_head.py:482-484)
            sv_l = {}
            for g, nm in enumerate(names):
                k = self._KEY[nm]
                _, lse = ops.ce_upsample_fwd(lgs[p][k], label, wgt, H, W, ii, acc=acc[g * P + p])
                sv_l[k] = (lgs[p][k], label, wgt, lse, H, W, n)
            sv_ls.append(sv_l)
        ckey = (tuple(names), tuple(sorted(coef.items())), str(dev))
        cvec = getattr(self, '_coef_vec', None)
        if cvec is None or cvec[0] != ckey:
            cvec = self._coef_vec = (ckey, torch.tensor([coef[self._KEY[nm]] for nm in names], dtype=torch.float32).to(dev))
        L = (acc[:, 0] * (lwt / n)).view(G, P)
        A = (acc[:, 1] * (100.0 / n)).view(G, P)
        ga = names.index('fusion') if 'fusion' in names else 0
        losses = [{'loss_seg': torch.dot(L[:, p], cvec[1]), 'acc_seg': A[ga, p:p + 1]} for p in range(P)]
        if P == 1:
            return losses[0], logits, (saved, sv_ls, coef)
        return losses, logits, (saved, sv_ls, coef)

    def bwd_train_joint(self, saved_all, B, gscale=None, mul=1.0):
        saved, sv_ls, coef = saved_all
        names, logits, P = saved[4], saved[5], saved[6]
        ii, lwt = self.ignore_index, self.loss_decode.loss_weight
        dl = torch.empty_like(logits)
        for g, n in enumerate(names):
            k = self._KEY[n]
            for p in range(P):
                ce_losses_bwd(sv_ls[p][k], gscale, mul * coef[k], ii, lwt, out=dl[(g * P + p) * B:(g * P + p + 1) * B])
        return self.bwd_joint(saved, dl, B)

    def fwd_train(self, inputs, B, gt, seg_weight=None, cfg=None):
        logits, saved = self.fwd(inputs, B)
        losses, sv_l, coef = self._loss_mix(logits, gt, seg_weight, cfg)
        return losses, logits, (saved, sv_l, coef)

    def bwd_train(self, saved_all, B, gscale=None, mul=1.0):
        saved, sv_l, coef = saved_all
        ii, lwt = self.ignore_index, self.loss_decode.loss_weight
        dlogits = {k: ce_losses_bwd(sv_l[k], gscale, mul * coef[k], ii, lwt) for k in sv_l}
        return self.bwd(saved, dlogits, B)

    def forward(self, inputs, cfg=None):
        """Reference signature: dict of NCHW feature lists -> dict of logits [B,nc,h,w] (no grad)."""
        B = inputs['f_image'][0].shape[0]
        feats = {k: (self._to_feats(v) if v is not None else None) for k, v in inputs.items()}
        out, _ = self.fwd(feats, B)
        return {k: (v.permute(0, 3, 1, 2) if v is not None else None) for k, v in out.items()}
