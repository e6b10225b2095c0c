"""Thin tensor-level wrappers over the C ABI (include/cmda_hip.h).  No autograd here: the
differentiable building blocks live in cmda_amd/functional.py and call these for both passes."""
import ctypes

import torch

from . import _lib as L
from ._lib import BF16, F32, GemmParams, View, c_f32, c_i32, c_i64, call, check_dev, dtype_tag, ptr, stream_of

_ESIZE = {torch.float32: 4, torch.bfloat16: 2}


def _chunk(t):
    return 16 // _ESIZE[t.dtype]


def plain_view(t, rows, cols, ld=None, batch_stride=0, offset=0):
    """View of a row-major matrix living inside tensor `t` (element `offset` from its start)."""
    ld = cols if ld is None else ld
    es = _ESIZE[t.dtype]
    ch = 16 // es
    base = t.data_ptr() + offset * es
    vec_ok = int(base % 16 == 0 and ld % ch == 0 and batch_stride % ch == 0)
    return View(ptr=base, ld=ld, R=rows, Cc=cols, batch_stride=batch_stride, conv=0, vec_ok=vec_ok,
                H=0, W=0, C=1, OH=1, OW=1, KH=1, KW=1, stride=1, pad=0, dil=1, in_dil=1, reflect=0)


def conv_view(x, B, H, W, C, KH, KW, stride, pad, dil=1, OH=None, OW=None, in_dil=1, reflect=0):
    """im2col view of NHWC tensor x[B,H,W,C]: r=(b,oh,ow), c=(kh,kw,ci)."""
    if OH is None:
        OH = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
        OW = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
    ch = _chunk(x)
    vec_ok = int(x.data_ptr() % 16 == 0 and C % ch == 0)
    return View(ptr=x.data_ptr(), ld=0, R=B * OH * OW, Cc=KH * KW * C, batch_stride=0, conv=1, H=H, W=W, C=C,
                OH=OH, OW=OW, KH=KH, KW=KW, stride=stride, pad=pad, dil=dil, in_dil=in_dil, reflect=reflect,
                vec_ok=vec_ok)


ACT = {None: 0, 'none': 0, 'relu': 1, 'gelu': 2}


def gemm(A, B, out, M, N, K, *, a_kstrided=False, b_kstrided=False, ldc=None, batch=1, c_batch_stride=0,
         splits=1, alpha=1.0, beta=0.0, bias=None, act=None, res=None, ldres=None, res_batch_stride=0,
         rowscale=None, rows_per_scale=1, atomic=False, dtype=None, c_offset=0):
    """out[m,n] = epi(alpha * sum_k A(m,k) B(n,k)); A/B are `View`s built by plain_view / conv_view."""
    check_dev(out, bias, res, rowscale)
    out_f32 = out.dtype == torch.float32
    p = GemmParams()
    p.A, p.B = A, B
    p.a_kstrided, p.b_kstrided = int(a_kstrided), int(b_kstrided)
    p.C = out.data_ptr() + c_offset * _ESIZE[out.dtype]
    p.ldc = N if ldc is None else ldc
    p.c_batch_stride = c_batch_stride
    p.M, p.N, p.K, p.batch, p.splits = M, N, K, batch, splits
    p.alpha, p.beta = alpha, beta
    p.bias = bias.data_ptr() if bias is not None else None
    p.act = ACT[act]
    p.res = res.data_ptr() if res is not None else None
    p.ldres = (N if ldres is None else ldres)
    p.res_batch_stride = res_batch_stride
    p.rowscale = rowscale.data_ptr() if rowscale is not None else None
    p.rows_per_scale = rows_per_scale
    p.dtype = dtype
    p.out_f32 = int(out_f32)
    assert out_f32 or not atomic
    p.atomic = int(atomic)
    call('cmda_gemm', ctypes.byref(p), stream_of(out))
    return out


def layernorm_fwd(x, gamma, beta, eps, save_stats=True):
    check_dev(x, gamma, beta)
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    call('cmda_layernorm_fwd', ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), c_i64(rows), c_i32(C),
         c_f32(eps), dtype_tag(x), stream_of(x))
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dres=None):
    check_dev(dy, x, gamma, mean, rstd, dgamma, dbeta, dres)
    C = x.shape[-1]
    rows = x.numel() // C
    dx = torch.empty_like(x)
    call('cmda_layernorm_bwd', ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dgamma),
         ptr(dbeta), c_i64(rows), c_i32(C), dtype_tag(x), stream_of(x))
    return dx
