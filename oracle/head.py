"""ORACLE (test infrastructure, not product code): CPU fp32 restatement of the DAFormer decode head and its loss.

Follows:
  decode_heads/daformer_head.py  ASPPWrapper :15-79, build_layer :82-116, DAFormerHead :136-197,
                                 DAFormerHeadFusion :200-322 (weight sharing :251-258)
  decode_heads/aspp_head.py :12-51, sep_aspp_head.py :12-27, segformer_head.py :18-28 (MLP)
  decode_heads/decode_head.py    BaseDecodeHead(.Fusion): cls_seg :563-586, losses :588-606, forward_train :423-534
  losses/cross_entropy_loss.py :11-34, losses/utils.py :48-77, losses/accuracy.py :6-51, ops/wrappers.py :9-28
mmcv 1.3.7's ConvModule / DepthwiseSeparableConvModule (third-party, not under /root/reference) are restated from
their documented behaviour: conv -> BN -> ReLU, conv bias dropped when a norm follows, sub-module names
conv / bn / activate and depthwise_conv / pointwise_conv; Kaiming-normal(fan_out, relu) conv init, BN weight 1 bias 0.
Parameter names match the reference.  Pinned by tests/golden (see oracle/mit.py header).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def resize(x, size):
    return F.interpolate(x, size=size, mode='bilinear', align_corners=False)


class ConvModule(nn.Module):
    def __init__(self, cin, cout, k, padding=0, dilation=1, groups=1, norm=True, act=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=padding, dilation=dilation, groups=groups, bias=not norm)
        if norm:
            self.bn = nn.BatchNorm2d(cout)
        self.with_norm, self.with_act = norm, act
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode='fan_out', nonlinearity='relu')
        if self.conv.bias is not None:
            nn.init.zeros_(self.conv.bias)

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            x = self.bn(x)
        return F.relu(x) if self.with_act else x


class DepthwiseSeparableConvModule(nn.Module):
    def __init__(self, cin, cout, k, padding, dilation):
        super().__init__()
        self.depthwise_conv = ConvModule(cin, cin, k, padding=padding, dilation=dilation, groups=cin)
        self.pointwise_conv = ConvModule(cin, cout, 1)

    def forward(self, x):
        return self.pointwise_conv(self.depthwise_conv(x))


class ASPPWrapper(nn.Module):
    def __init__(self, in_channels, channels, sep, dilations, **_):
        super().__init__()
        mods = []
        for d in dilations:
            if d == 1:
                mods.append(ConvModule(in_channels, channels, 1))
            elif sep:
                mods.append(DepthwiseSeparableConvModule(in_channels, channels, 3, padding=d, dilation=d))
            else:
                mods.append(ConvModule(in_channels, channels, 3, padding=d, dilation=d))
        self.aspp_modules = nn.ModuleList(mods)
        self.bottleneck = ConvModule(len(dilations) * channels, channels, 3, padding=1)

    def forward(self, x):
        return self.bottleneck(torch.cat([m(x) for m in self.aspp_modules], dim=1))


class MLP(nn.Module):
    def __init__(self, input_dim, embed_dim):
        super().__init__()
        self.proj = nn.Linear(input_dim, embed_dim)

    def forward(self, x):
        return self.proj(x.flatten(2).transpose(1, 2))


def build_layer(cin, cout, type, **kw):
    if type == 'mlp':
        return MLP(cin, cout)
    if type == 'aspp':
        return ASPPWrapper(cin, cout, kw.get('sep', True), kw.get('dilations', (1, 6, 12, 18)))
    if type == 'conv':
        return ConvModule(cin, cout, kw['kernel_size'], padding=kw['kernel_size'] // 2)
    raise NotImplementedError(type)


def cross_entropy_loss(seg_logit, seg_label, seg_weight=None, ignore_index=255, loss_weight=1.0):
    """losses(): up-sample logits to the label size, CE 'none', x weight, mean over ALL pixels; top-1 accuracy in %."""
    seg_logit = resize(seg_logit, seg_label.shape[2:])
    lab = seg_label.squeeze(1)
    loss = F.cross_entropy(seg_logit, lab, reduction='none', ignore_index=ignore_index)
    if seg_weight is not None:
        loss = loss * seg_weight.float()
    acc = (seg_logit.argmax(1) == lab).float().sum()[None] * (100.0 / lab.numel())  # shape (1,), as accuracy() returns
    return {'loss_seg': loss_weight * loss.mean(), 'acc_seg': acc}


FUSION_CFG = dict(type='aspp', sep=True, dilations=(1, 6, 12, 18))


class _HeadBase(nn.Module):
    def __init__(self, in_channels=(64, 128, 320, 512), channels=256, num_classes=19, dropout_ratio=0.1,
                 embed_dims=256, fusion_cfg=None, ignore_index=255):
        super().__init__()
        self.in_channels, self.channels, self.num_classes = in_channels, channels, num_classes
        self.ignore_index = ignore_index
        self.embed_dims = [embed_dims] * len(in_channels) if isinstance(embed_dims, int) else list(embed_dims)
        self.fusion_cfg = dict(FUSION_CFG if fusion_cfg is None else fusion_cfg)
        self.conv_seg = nn.Conv2d(channels, num_classes, 1)
        nn.init.normal_(self.conv_seg.weight, std=0.01)
        nn.init.zeros_(self.conv_seg.bias)
        self.dropout = nn.Dropout2d(dropout_ratio) if dropout_ratio > 0 else None

    def _make_branch(self):
        embeds = nn.ModuleDict({str(i): MLP(c, e) for i, (c, e) in enumerate(zip(self.in_channels, self.embed_dims))})
        fuse = build_layer(sum(self.embed_dims), self.channels, **self.fusion_cfg)
        return embeds, fuse

    @staticmethod
    def _branch(embeds, fuse, feats):
        n = feats[-1].shape[0]
        os_size = feats[0].shape[2:]
        cs = []
        for i, f in enumerate(feats):
            c = embeds[str(i)](f).permute(0, 2, 1).reshape(n, -1, f.shape[2], f.shape[3])
            if c.shape[2:] != os_size:
                c = resize(c, os_size)
            cs.append(c)
        return fuse(torch.cat(cs, dim=1))

    def cls_seg(self, feat, with_dropout=True):
        if with_dropout and self.dropout is not None:
            feat = self.dropout(feat)
        return self.conv_seg(feat)

    def losses(self, seg_logit, seg_label, seg_weight=None):
        return cross_entropy_loss(seg_logit, seg_label, seg_weight, self.ignore_index)


class DAFormerHead(_HeadBase):
    def __init__(self, **kw):
        super().__init__(**kw)
        self.embed_layers, self.fuse_layer = self._make_branch()

    def forward(self, feats):
        return self.cls_seg(self._branch(self.embed_layers, self.fuse_layer, feats))

    def forward_train(self, feats, gt, seg_weight=None):
        logits = self.forward(feats)
        return self.losses(logits, gt, seg_weight), logits


class DAFormerHeadFusion(_HeadBase):
    def __init__(self, share_decoder=True, **kw):
        super().__init__(**kw)
        self.embed_layers_image, self.fuse_layer_image = self._make_branch()
        self.embed_layers_events, self.fuse_layer_events = self._make_branch()
        self.embed_layers_fusion, self.fuse_layer_fusion = self._make_branch()
        if share_decoder:
            self.embed_layers_events = self.embed_layers_fusion = self.embed_layers_image
            self.fuse_layer_events = self.fuse_layer_fusion = self.fuse_layer_image

    def forward(self, inputs, cfg=None):
        out = {'events_output': None, 'fusion_output': None, 'img_self_res_output': None}
        out['image_output'] = self.cls_seg(self._branch(self.embed_layers_image, self.fuse_layer_image, inputs['f_image']))
        if inputs.get('f_events') is not None:
            out['events_output'] = self.cls_seg(
                self._branch(self.embed_layers_events, self.fuse_layer_events, inputs['f_events']), with_dropout=False)
        if inputs.get('f_fusion') is not None:
            out['fusion_output'] = self.cls_seg(
                self._branch(self.embed_layers_fusion, self.fuse_layer_fusion, inputs['f_fusion']), with_dropout=False)
        if inputs.get('f_img_self_res') is not None:
            out['img_self_res_output'] = self.cls_seg(
                self._branch(self.embed_layers_events, self.fuse_layer_events, inputs['f_img_self_res']), with_dropout=False)
        return out

    def forward_train(self, inputs, gt, seg_weight=None, cfg=None):
        """decode_head.py:423-534, the non-split / non-confidence branch used by configs/fusion/*."""
        lw = cfg['loss_weight']
        logits = self.forward(inputs, cfg)
        if seg_weight is None:
            seg_weight = torch.ones_like(gt)[:, 0]
        l_img = self.losses(logits['image_output'], gt, seg_weight)
        l_evt = self.losses(logits['events_output'], gt, seg_weight)
        losses = {}
        if logits['fusion_output'] is not None:
            l_fus = self.losses(logits['fusion_output'], gt, seg_weight)
            losses['loss_seg'] = l_fus['loss_seg'] * lw['fusion'] + l_img['loss_seg'] * lw['image']
        else:
            losses['loss_seg'] = l_img['loss_seg'] * lw['image']
        if logits['img_self_res_output'] is not None:
            l_isr = self.losses(logits['img_self_res_output'], gt, seg_weight)
            losses['loss_seg'] = losses['loss_seg'] + (l_isr['loss_seg'] * lw['img_self_res'] + l_evt['loss_seg'] * (lw['events'] / 2))
        else:
            losses['loss_seg'] = losses['loss_seg'] + l_evt['loss_seg'] * lw['events']
        losses['acc_seg'] = l_fus['acc_seg'] if logits['fusion_output'] is not None else l_img['acc_seg']
        return losses, logits
