"""Image Motion-Extractor generator (CycleGAN ResNet-9 blocks, inference only) on the HIP kernels.

Mirrors mmseg/models/cyclegan/cyclegan_model.py: define_G :119-160, ResnetGenerator :316-374, ResnetBlock :377-434, as
DACS uses it (uda/dacs.py:96-103,400-404: define_G() = 1->1 channels, ngf 64, InstanceNorm2d(affine=False), reflect
padding, 9 blocks, run frozen under no_grad on mean_c(img_time_res)).  Parameter names are the reference's
`model.<idx>...` so `cityscapes_ICD_to_dsec_EN.pth` loads with load_state_dict.

Every convolution is an implicit GEMM on the MFMA kernel (reflection padding and the transposed convolutions'
zero insertion are address modes of the operand view, gemm.hip view_offset); InstanceNorm reuses the column-statistics
kernels of batchnorm.hip per sample.
"""
import torch
import torch.nn as nn

from . import nn as K
from . import ops
from . import runtime as rt
from .ops import conv_view, plain_view


class ResnetBlock(nn.Module):
    def __init__(self, dim, padding_type='reflect', norm_layer=None, use_dropout=False, use_bias=True):
        super().__init__()
        assert padding_type == 'reflect' and not use_dropout
        self.conv_block = nn.Sequential(nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, 3, bias=use_bias),
                                        nn.InstanceNorm2d(dim), nn.ReLU(True), nn.ReflectionPad2d(1),
                                        nn.Conv2d(dim, dim, 3, bias=use_bias), nn.InstanceNorm2d(dim))


class ResnetGenerator(nn.Module):
    def __init__(self, input_nc=1, output_nc=1, ngf=64, norm_layer=None, use_dropout=False, n_blocks=9,
                 padding_type='reflect'):
        super().__init__()
        m = [nn.ReflectionPad2d(3), nn.Conv2d(input_nc, ngf, 7, bias=True), nn.InstanceNorm2d(ngf), nn.ReLU(True)]
        for i in range(2):
            c = ngf * 2 ** i
            m += [nn.Conv2d(c, c * 2, 3, stride=2, padding=1, bias=True), nn.InstanceNorm2d(c * 2), nn.ReLU(True)]
        m += [ResnetBlock(ngf * 4) for _ in range(n_blocks)]
        for i in range(2):
            c = ngf * 2 ** (2 - i)
            m += [nn.ConvTranspose2d(c, c // 2, 3, stride=2, padding=1, output_padding=1, bias=True),
                  nn.InstanceNorm2d(c // 2), nn.ReLU(True)]
        m += [nn.ReflectionPad2d(3), nn.Conv2d(ngf, output_nc, 7), nn.Tanh()]
        self.model = nn.Sequential(*m)
        self.n_blocks = n_blocks
        self._ident = {}

    def _inorm(self, x, B, HW, C, relu, res=None, out_dtype=None, copy=False, stats=None):
        """InstanceNorm2d(affine=False, eps 1e-5) (+ReLU) (+ fp32 residual) on NHWC-flat x [B*HW, C]: the grouped column-statistics
        kernels with one group per sample (up to 8 samples per launch).  x is the convolution's fp32 output in both modes; the
        result is stored as `out_dtype` (default: the compute dtype, the next convolution's operand).  copy=True (the residual
        stream, kept fp32: out_dtype = float32): also returns a compute-dtype copy for the next convolution, written by the same
        launch in the bf16 mode."""
        key = (C, str(x.device))
        ident = self._ident.get(key)
        if ident is None:
            ident = self._ident[key] = (torch.ones(C, dtype=torch.float32, device=x.device),
                                        torch.zeros(C, dtype=torch.float32, device=x.device))
        one, zero = ident
        cd = rt.compute_dtype()
        y = torch.empty(x.shape, dtype=out_dtype or cd, device=x.device)
        y2 = torch.empty(x.shape, dtype=cd, device=x.device) if (copy and y.dtype != cd) else None
        for b0 in range(0, B, 8):
            g = min(8, B - b0)
            sl = slice(b0 * HW, (b0 + g) * HW)
            ops.bn_train_fwd2(x[sl], one, zero, y[sl], HW, C, 1e-5, relu, groups=g, res32=None if res is None else res[sl],
                              y2=None if y2 is None else y2[sl], stats_ws=stats if B <= 8 else None)
        return (y, y if y2 is None else y2) if copy else y

    @staticmethod
    def _stats_ws(dev, B, HW, C):
        """the workspace the convolution's epilogue leaves the InstanceNorm statistics in (one group per sample), or None"""
        return ops.bn_stats_ws(dev, B, C) if (B <= 8 and ops.colstats_ok(HW, C)) else None

    def _conv(self, x, conv, B, H, W, stride, pad, reflect, act=None, stats=False):
        """fp32 output in both modes: the InstanceNorm behind every convolution takes its statistics from the unrounded sums.
        stats=True: ... in the epilogue of this very launch where the shapes allow; returns (y, OH, OW, workspace or None)"""
        Co, _, KH, _ = conv.weight.shape
        OH, OW = K.conv_out_size(H, W, KH, stride, pad)
        out = torch.empty(B * OH * OW, Co, dtype=torch.float32, device=x.device)
        ws = self._stats_ws(x.device, B, OH * OW, Co) if stats else None
        y, OH, OW = K.conv_fwd(x, conv.weight, conv.bias, B, H, W, stride, pad, 1, act=act, reflect=reflect, out=out,
                               colstats=None if ws is None else (ws, OH * OW))
        return (y, OH, OW, ws) if stats else (y, OH, OW)

    def _convT(self, x, ct, B, H, W):
        """ConvTranspose2d(k3, s2, p1, output_padding 1) = conv of the zero-inserted input with the flipped kernel."""
        Ci, Co, KH, KW = ct.weight.shape
        OH, OW = 2 * H, 2 * W
        y = torch.empty(B * OH * OW, Co, dtype=torch.float32, device=x.device)
        ws = self._stats_ws(x.device, B, OH * OW, Co)
        ops.gemm(conv_view(x, B, H, W, Ci, KH, KW, 1, KH - 1 - 1, 1, OH=OH, OW=OW, in_dil=2),
                 plain_view(rt.wconv(ct.weight, 'dgrad'), Co, KH * KW * Ci), y, B * OH * OW, Co, KH * KW * Ci,
                 dtype=rt.tag(), bias=ct.bias, colstats=None if ws is None else (ws, OH * OW))
        return y, OH, OW, ws

    @torch.no_grad()
    def forward_mean3(self, img_time_res):
        """dacs.py:400-404: G(mean_c(img_time_res)) repeated to 3 channels; [B,3,H,W] -> [B,3,H,W]"""
        return self.forward(img_time_res.mean(dim=1, keepdim=True)).repeat(1, 3, 1, 1)

    @ops.sited('generator')
    @torch.no_grad()
    def forward(self, inp):
        """inp fp32 NCHW [B,1,H,W] -> fp32 NCHW [B,1,H,W] (tanh)."""
        B, Cin, H, W = inp.shape
        m = self.model
        cp = rt.conv_channel_pad(Cin)   # bf16 mode: the 1-channel input padded to 8 (16-byte im2col chunks: the LDS-DMA GEMM path)
        x = torch.empty(B * H * W, cp, dtype=rt.compute_dtype(), device=inp.device)
        if cp != Cin:
            ops.nchw_to_nhwc_pad(inp.contiguous(), x, B, Cin, H * W, cp)
            H1, W1 = K.conv_out_size(H + 6, W + 6, 7, 1, 0)
            c = torch.empty(B * H1 * W1, m[1].out_channels, dtype=torch.float32, device=inp.device)
            ws = self._stats_ws(inp.device, B, H1 * W1, m[1].out_channels)
            K.conv_fwd(x, m[1].weight, m[1].bias, B, H, W, 1, 3, 1, reflect=1, ci_pad=cp, out=c,
                       colstats=None if ws is None else (ws, H1 * W1))
        else:
            ops.permute4(inp.contiguous(), x, (B, Cin, H, W), (0, 2, 3, 1))
            c, H1, W1, ws = self._conv(x, m[1], B, H, W, 1, 3, 1, stats=True)
        # every InstanceNorm below takes its statistics from the epilogue of the convolution in front of it (`ws`)
        x = self._inorm(c, B, H1 * W1, m[1].out_channels, True, stats=ws)
        c, H2, W2, ws = self._conv(x, m[4], B, H1, W1, 2, 1, 0, stats=True)
        x = self._inorm(c, B, H2 * W2, m[4].out_channels, True, stats=ws)
        c, H3, W3, ws = self._conv(x, m[7], B, H2, W2, 2, 1, 0, stats=True)
        C = m[7].out_channels
        # the residual stream of the nine ResNet blocks stays fp32 (xs); xb = its compute-dtype copy, the convolutions' operand
        xs, xb = self._inorm(c, B, H3 * W3, C, True, out_dtype=torch.float32, copy=True, stats=ws)
        for i in range(self.n_blocks):
            cb = m[10 + i].conv_block
            c, _, _, ws = self._conv(xb, cb[1], B, H3, W3, 1, 1, 1, stats=True)
            y = self._inorm(c, B, H3 * W3, C, True, stats=ws)
            c, _, _, ws = self._conv(y, cb[5], B, H3, W3, 1, 1, 1, stats=True)
            xs, xb = self._inorm(c, B, H3 * W3, C, False, res=xs, out_dtype=torch.float32, copy=True, stats=ws)
        x = xb
        k = 10 + self.n_blocks
        c, H4, W4, ws = self._convT(x, m[k], B, H3, W3)
        x = self._inorm(c, B, H4 * W4, m[k].out_channels, True, stats=ws)
        c, H5, W5, ws = self._convT(x, m[k + 3], B, H4, W4)
        x = self._inorm(c, B, H5 * W5, m[k + 3].out_channels, True, stats=ws)
        last = m[k + 7]
        Co = last.out_channels
        Ci = last.in_channels
        if Co == 1 and ops.conv_co1_ok(x, Ci, 7, 3):   # one output channel: a stencil kernel, not an N = 1 GEMM
            y = ops.conv_co1(x, rt.wconv(last.weight), last.bias, B, H5, W5, Ci, 7, 3, True, 'tanh')
            return y.view(B, 1, H5, W5)
        y = torch.empty(B * H5 * W5, Co, dtype=torch.float32, device=inp.device)
        ops.gemm(conv_view(x, B, H5, W5, Ci, 7, 7, 1, 3, 1, OH=H5, OW=W5, reflect=1),
                 plain_view(rt.wconv(last.weight), Co, 49 * Ci), y, B * H5 * W5, Co, 49 * Ci, dtype=rt.tag(),
                 bias=last.bias, act='tanh')
        return y.view(B, H5, W5, Co).permute(0, 3, 1, 2)


def define_G(input_nc=1, output_nc=1, ngf=64, netG='resnet_9blocks', norm='instance', use_dropout=False,
             init_type='normal', init_gain=0.02, gpu_ids=[]):
    assert netG in ('resnet_9blocks', 'resnet_6blocks') and norm == 'instance' and not use_dropout
    net = ResnetGenerator(input_nc, output_nc, ngf, n_blocks=9 if netG == 'resnet_9blocks' else 6)
    for mod in net.modules():
        if isinstance(mod, (nn.Conv2d, nn.ConvTranspose2d)):
            nn.init.normal_(mod.weight.data, 0.0, init_gain)
            if mod.bias is not None:
                nn.init.constant_(mod.bias.data, 0.0)
    return net
