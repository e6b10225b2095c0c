"""Hand-scheduled forward/backward composites over the HIP kernels (no torch autograd inside).

Activations are 2-D [tokens, channels] tensors (NLC / NHWC flattened) in the compute dtype.  Every `*_fwd` returns
its outputs plus whatever the matching `*_bwd` needs; parameter gradients are accumulated by the kernels into
`param.grad` (fp32) through cmda_amd.runtime.grad().
"""
import os

import torch

from . import ops
from . import runtime as rt
from .ops import conv_view, plain_view


# ------------------------------------------------------------------ Linear
def linear_fwd(x, weight, bias, M, K, *, act=None, res=None, rowscale=None, rows_per_scale=1, out=None, ldc=None,
               c_offset=0, out_dtype=None, x_ld=None, x_off=0, colstats=None):
    """y[M,N] = act(x[M,K] @ W[N,K]^T + b) (* rowscale) (+ res); colstats: ops.gemm"""
    N = weight.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype or rt.compute_dtype(), device=x.device)
    ops.gemm(plain_view(x, M, K, ld=x_ld, offset=x_off), plain_view(rt.w(weight), N, K), out, M, N, K, dtype=rt.tag(),
             bias=bias, act=act, res=res, rowscale=rowscale, rows_per_scale=rows_per_scale, ldc=ldc, c_offset=c_offset, colstats=colstats)
    return out


def linear_bwd(dy, x, weight, bias, M, K, *, need_dx=True, dx_out=None, dx_beta=0.0, dy_ld=None, dy_off=0, x_ld=None,
               x_off=0):
    """dW += dy^T x, db += colsum(dy), returns dx = dy @ W (optionally accumulated into dx_out)"""
    N = weight.shape[0]
    dyv_k = plain_view(dy, M, N, ld=dy_ld, offset=dy_off)  # (r = token, c = n)
    xv = plain_view(x, M, K, ld=x_ld, offset=x_off)
    fused = (bias is not None and rt.tag() == 1 and dyv_k.vec_ok and xv.vec_ok and N % 8 == 0 and K % 8 == 0
             and (dy_ld or N) % 8 == 0 and (x_ld or K) % 8 == 0)
    # split-bf16 mode: the lean weight-gradient kernel (csrc/gemm_x3_lean.hip) folds the bias gradient as well -- its eligibility
    # (cmda_gemm_x3_lean_ok_): plain fp32 operands, token count a multiple of the 32-deep k-tile, 16-byte rows
    fused = fused or (bias is not None and rt.tag() == 2 and dyv_k.vec_ok and xv.vec_ok and M % 32 == 0 and N % 4 == 0 and K % 4 == 0
                      and (dy_ld or N) % 4 == 0 and (x_ld or K) % 4 == 0 and ops.GEMM_TILE_HINT == 0)
    def wgrad():   # off the critical dgrad chain when a block-level batch is open (runtime.lane_batch)
        ops.gemm(dyv_k, xv, rt.grad(weight), N, K, M, a_kstrided=True, b_kstrided=True, dtype=rt.tag(), atomic=True,
                 splits=0, colsum=rt.grad(bias) if fused else None,   # bias gradient rides along in the wgrad kernel
                 defer=True, keep=(dy, x))   # inside a backward pass: queued, launched in groups (ops.gemm_flush_deferred)
        if bias is not None and not fused:
            ops.colsum(dy, rt.grad(bias), M, N, ld=dy_ld, offset=dy_off)
    rt.side('wgrad', wgrad, dy, x)
    if not need_dx:
        return None
    dx = dx_out if dx_out is not None else torch.empty(M, K, dtype=rt.compute_dtype(), device=dy.device)
    # (round 6: the same product through a TRANSPOSED bf16 weight mirror -- both operands K-contiguous, the forward's kernel form --
    # measured 1.0 ms per step SLOWER, same box, three alternating runs: 55.5 against 56.5-56.7 ms; the ~155 M Linear weights re-transposed
    # per step cost more than the K-strided k-tile loses inside the two-lane step.  Removed)
    ops.gemm(dyv_k, plain_view(rt.w(weight), N, K), dx, M, K, N, b_kstrided=True, dtype=rt.tag(), beta=dx_beta)
    return dx


# ------------------------------------------------------------------ Conv2d as implicit GEMM (NHWC)
def conv_out_size(H, W, k, stride, pad, dil=1):
    return (H + 2 * pad - dil * (k - 1) - 1) // stride + 1, (W + 2 * pad - dil * (k - 1) - 1) // stride + 1


def conv_fwd(x, weight, bias, B, H, W, stride, pad, dil=1, *, act=None, out=None, reflect=0, ci_pad=0, colstats=None):
    """ci_pad > Ci: x carries ci_pad channels per pixel (zeros past Ci) and the weight copy is padded alike (runtime.wconv)"""
    Co, Ci, KH, KW = weight.shape
    Ci = max(Ci, ci_pad)
    OH, OW = conv_out_size(H, W, KH, stride, pad, dil)
    M, K = B * OH * OW, KH * KW * Ci
    if out is None:
        out = torch.empty(M, Co, dtype=rt.compute_dtype(), device=x.device)
    ops.gemm(conv_view(x, B, H, W, Ci, KH, KW, stride, pad, dil, OH=OH, OW=OW, reflect=reflect),
             plain_view(rt.wconv(weight, ci_pad=ci_pad), Co, K), out, M, Co, K, dtype=rt.tag(), bias=bias, act=act, colstats=colstats)
    return out, OH, OW


def conv_bwd(dy, x, weight, bias, B, H, W, stride, pad, dil=1, *, need_dx=True, dx_out=None, dx_beta=0.0, ci_pad=0):
    Co, Ci, KH, KW = weight.shape
    if ci_pad > Ci:   # channel-padded input (the encoder's first convolution): weight gradient through a padded shadow, no dx
        assert not need_dx
        with ops.ln_deferral():   # (a scope of its own when called outside a backward pass: the shadow is drained on exit)
            return _conv_bwd(dy, x, weight, bias, B, H, W, stride, pad, dil, need_dx=False, ci_pad=ci_pad)
    return _conv_bwd(dy, x, weight, bias, B, H, W, stride, pad, dil, need_dx=need_dx, dx_out=dx_out, dx_beta=dx_beta)


def _conv_bwd(dy, x, weight, bias, B, H, W, stride, pad, dil=1, *, need_dx=True, dx_out=None, dx_beta=0.0, ci_pad=0):
    Co, Ci, KH, KW = weight.shape
    Ci = max(Ci, ci_pad)
    OH, OW = conv_out_size(H, W, KH, stride, pad, dil)
    M, K = B * OH * OW, KH * KW * Ci
    def wgrad():
        # dW[co, (kh,kw,ci)] accumulated with atomics; outside a deferral scope straight into the parameter's [Co,Ci,KH,KW]
        # gradient (c_perm: no staging buffer); the bias gradient rides along in the same kernel when the operands allow
        dyv = plain_view(dy, M, Co)
        fused = (bias is not None and rt.tag() == 1 and dyv.vec_ok and Co % 8 == 0 and K % 8 == 0 and Ci % 8 == 0)
        shadow = ops.conv_grad_shadow(rt.grad(weight), ci_pad)
        if shadow is not None:   # inside a pass: coalesced atomics into the [Co,KH,KW,Ci] shadow, drained once per pass
            ops.gemm(dyv, conv_view(x, B, H, W, Ci, KH, KW, stride, pad, dil, OH=OH, OW=OW), shadow, Co, K, M,
                     a_kstrided=True, b_kstrided=True, dtype=rt.tag(), atomic=True, splits=0,
                     colsum=rt.grad(bias) if fused else None, defer=True, keep=(dy, x))
        else:
            ops.gemm(dyv, conv_view(x, B, H, W, Ci, KH, KW, stride, pad, dil, OH=OH, OW=OW), rt.grad(weight), Co, K, M,
                     a_kstrided=True, b_kstrided=True, dtype=rt.tag(), atomic=True, splits=0, c_perm=(Ci, KH * KW),
                     colsum=rt.grad(bias) if fused else None)
        if bias is not None and not fused:
            ops.colsum(dy, rt.grad(bias), M, Co)
    rt.side('wgrad', wgrad, dy, x)
    if not need_dx:
        return None
    dx = dx_out if dx_out is not None else torch.empty(B * H * W, Ci, dtype=rt.compute_dtype(), device=dy.device)
    if KH == stride and pad == 0 and dil == 1 and H % stride == 0 and W % stride == 0:
        # non-overlapping patches (spatial-reduction conv): dcol = dy @ W, then un-patchify (pure permutation)
        # the GEMM stores each row (b,oh,ow) x column (kh,kw,ci) straight at its NHWC position (c_patch): no column buffer
        ops.gemm(plain_view(dy, M, Co), plain_view(rt.wconv(weight), Co, K), dx, M, K, Co, b_kstrided=True, dtype=rt.tag(),
                 beta=dx_beta, c_patch=(OW, KH, KW * Ci))
        return dx
    ops.gemm(conv_view(dy, B, OH, OW, Co, KH, KW, 1, dil * (KH - 1) - pad, dil, OH=H, OW=W, in_dil=stride),
             plain_view(rt.wconv(weight, 'dgrad'), Ci, KH * KW * Co), dx, B * H * W, Ci, KH * KW * Co, dtype=rt.tag(),
             beta=dx_beta)
    return dx


# ------------------------------------------------------------------ attention (scores materialised; v0 path)
def attention_fwd(q, kv, B, N, Nk, heads, C, scale, need_grad=True):
    """q [B*N,C], kv [B*Nk,2C] -> o [B*N,C]; returns (o, P): P [B,heads,N,Nk] saved for the backward, or None when the fused
    kernel ran (bf16 -- or fp32 storage in the split-bf16 mode --, head_dim 64, Nk <= 256: the backward recomputes the probabilities in LDS)."""
    if ops.attention_fused_ok(q, Nk, heads, C, need_grad, x3=rt.gemm_x3()) and not os.environ.get('CMDA_NO_FUSED_ATTENTION'):
        return ops.attention_fused_fwd(q, kv, B, N, Nk, heads, C, scale), None
    hd = C // heads
    dev = q.device
    P = torch.empty(B, heads, N, Nk, dtype=rt.compute_dtype(), device=dev)
    ops.gemm(plain_view(q, N, hd, ld=C, batch_stride=N * C, batch2_stride=hd),
             plain_view(kv, Nk, hd, ld=2 * C, batch_stride=Nk * 2 * C, batch2_stride=hd),
             P, N, Nk, hd, batch=B, batch2=heads, c_batch_stride=heads * N * Nk, c_batch2_stride=N * Nk, dtype=rt.tag())
    ops.softmax_fwd_(P, B * heads * N, Nk, scale)
    o = torch.empty(B * N, C, dtype=rt.compute_dtype(), device=dev)
    ops.gemm(plain_view(P, N, Nk, batch_stride=heads * N * Nk, batch2_stride=N * Nk),
             plain_view(kv, Nk, hd, ld=2 * C, batch_stride=Nk * 2 * C, batch2_stride=hd, offset=C),
             o, N, hd, Nk, b_kstrided=True, batch=B, batch2=heads, ldc=C, c_batch_stride=N * C, c_batch2_stride=hd,
             dtype=rt.tag())
    return o, P


def attention_bwd(do, q, kv, P, B, N, Nk, heads, C, scale):
    """returns (dq [B*N,C], dkv [B*Nk,2C]) in the compute dtype."""
    hd = C // heads
    dev = do.device
    tag = rt.tag()
    if P is None:  # fused forward ran (bf16)
        if ops.attention_bwd_direct(B, N, Nk, heads):   # few queries: dK | dV come out final, as bf16, from one block per key slice
            dkv = torch.empty(B * Nk, 2 * C, dtype=rt.compute_dtype(), device=dev)
            dq = ops.attention_fused_bwd(q, kv, do, None, B, N, Nk, heads, C, scale, dkv16=dkv)
            return dq, dkv
        # otherwise dK | dV accumulate in the persistent zeroed workspace, drained by one cast+clear
        dkv32 = ops.zero_ws(dev, B * Nk * 2 * C).view(B * Nk, 2 * C)
        dq = ops.attention_fused_bwd(q, kv, do, dkv32, B, N, Nk, heads, C, scale)
        return dq, ops.cast_clear(dkv32, rt.compute_dtype())
    dkv32 = torch.zeros(B * Nk, 2 * C, dtype=torch.float32, device=dev)
    Pv = dict(batch_stride=heads * N * Nk, batch2_stride=N * Nk)
    sp = 0  # auto split-K
    # dV_h = P_h^T dO_h
    ops.gemm(plain_view(P, N, Nk, **Pv), plain_view(do, N, hd, ld=C, batch_stride=N * C, batch2_stride=hd),
             dkv32, Nk, hd, N, a_kstrided=True, b_kstrided=True, batch=B, batch2=heads, ldc=2 * C,
             c_batch_stride=Nk * 2 * C, c_batch2_stride=hd, c_offset=C, dtype=tag, atomic=True, splits=sp)
    # dP_h = dO_h V_h^T
    dP = torch.empty_like(P)
    ops.gemm(plain_view(do, N, hd, ld=C, batch_stride=N * C, batch2_stride=hd),
             plain_view(kv, Nk, hd, ld=2 * C, batch_stride=Nk * 2 * C, batch2_stride=hd, offset=C),
             dP, N, Nk, hd, batch=B, batch2=heads, c_batch_stride=heads * N * Nk, c_batch2_stride=N * Nk, dtype=tag)
    ops.softmax_bwd_(P, dP, B * heads * N, Nk, scale)  # dP now holds dS (scale folded in)
    # dQ_h = dS_h K_h
    dq = torch.empty(B * N, C, dtype=rt.compute_dtype(), device=dev)
    ops.gemm(plain_view(dP, N, Nk, **Pv), plain_view(kv, Nk, hd, ld=2 * C, batch_stride=Nk * 2 * C, batch2_stride=hd),
             dq, N, hd, Nk, b_kstrided=True, batch=B, batch2=heads, ldc=C, c_batch_stride=N * C, c_batch2_stride=hd,
             dtype=tag)
    # dK_h = dS_h^T Q_h
    ops.gemm(plain_view(dP, N, Nk, **Pv), plain_view(q, N, hd, ld=C, batch_stride=N * C, batch2_stride=hd),
             dkv32, Nk, hd, N, a_kstrided=True, b_kstrided=True, batch=B, batch2=heads, ldc=2 * C,
             c_batch_stride=Nk * 2 * C, c_batch2_stride=hd, dtype=tag, atomic=True, splits=sp)
    dkv = dkv32 if rt.compute_dtype() == torch.float32 else ops.cast(dkv32, rt.compute_dtype())
    return dq, dkv


# split K of a spatial-reduction convolution when its output has at most this many 64 x 64 tiles (block_fwd).  32: the teacher's
# stage-3 convolution (512 rows x 320 channels, K = 1280: 40 tiles) stays one launch of the lean kernel instead of rows_fill + a
# split-K grid (59.0 -> 58.5 ms, gpurun r04y; 16 / 0 measured the same as 32)
SR_SPLITK_TILES = int(os.environ.get('CMDA_SR_SPLITK_TILES', '32'))


# ------------------------------------------------------------------ MiT Block (mix_transformer.py:108-148)
def block_fwd(x, p, B, H, W, C, heads, sr, *, eps=1e-6, dp1=None, dp2=None, save=True):
    """x [B*H*W, C].  p: the Block module (norm1, attn.{q,kv,proj,sr,norm}, norm2, mlp.{fc1,dwconv.dwconv,fc2}).
    dp1/dp2: per-sample DropPath scales (fp32 [B]) or None."""
    N = H * W
    M = B * N
    a = p.attn
    cd = rt.compute_dtype()
    sd = x.dtype    # residual-stream storage: fp32 in the bf16 mode's fp32-stream option (runtime.residual_fp32), else the compute dtype
    # (round 4's two-problem launches (q + sr-conv) and the LayerNorm-prologue Linear measured no faster than the eight-wave lean kernel's
    # separate launches on the step and were removed in round 5: DESIGN.md section 5)
    xn, m1, r1 = ops.layernorm_fwd(x, p.norm1.weight, p.norm1.bias, eps, out_dtype=cd)
    q = linear_fwd(xn, a.q.weight, a.q.bias, M, C)
    if sr > 1:
        OH, OW = conv_out_size(H, W, sr, sr, 0)
        Ksr = sr * sr * C
        # (split-bf16 mode as well, and there for every spatial-reduction convolution of few tiles: 512 x 320 x 1280 ran 80 us as one
        # launch of 40 workgroups -- 6.6 ms per step over the teacher's 82 calls)
        splitk = (rt.tag() in (1, 2) and Ksr >= 1024 and C % 4 == 0 and
                  ((B * OH * OW + 63) // 64) * ((C + 63) // 64) <= (SR_SPLITK_TILES if rt.tag() == 1 else 3 * SR_SPLITK_TILES))
        if splitk:
            # few output tiles, long contraction (stages 1 / 2: B*256 rows x 64 / 128 channels over K = 4096 / 2048): 8 - 32 workgroups
            # running 32 - 64 k-tiles one after the other (39 / 23 us).  Split K instead: the slices accumulate with fp32 atomics on
            # top of the bias, and the LayerNorm behind reads the fp32 sums (which it also keeps for its backward pass).
            xs_pre = ops.rows_fill(torch.empty(B * OH * OW, C, dtype=torch.float32, device=x.device), a.sr.bias)
            ops.gemm(conv_view(xn, B, H, W, C, sr, sr, sr, 0, 1, OH=OH, OW=OW), plain_view(rt.wconv(a.sr.weight), C, Ksr), xs_pre,
                     B * OH * OW, C, Ksr, dtype=rt.tag(), atomic=True, splits=0)
        else:
            xs_pre, OH, OW = conv_fwd(xn, a.sr.weight, a.sr.bias, B, H, W, sr, 0)
        Nk = OH * OW
        xs, ms, rs = ops.layernorm_fwd(xs_pre, a.norm.weight, a.norm.bias, 1e-5, out_dtype=cd)
        kv = linear_fwd(xs, a.kv.weight, a.kv.bias, B * Nk, C)
    else:
        xs_pre, ms, rs, xs, Nk = None, None, None, xn, N
        kv = linear_fwd(xs, a.kv.weight, a.kv.bias, B * Nk, C)
    hd = C // heads
    o, P = attention_fwd(q, kv, B, N, Nk, heads, C, hd ** -0.5, need_grad=save)
    x1 = linear_fwd(o, a.proj.weight, a.proj.bias, M, C, res=x, rowscale=dp1, rows_per_scale=N, out_dtype=sd)
    xn2, m2, r2 = ops.layernorm_fwd(x1, p.norm2.weight, p.norm2.bias, eps, out_dtype=cd)
    hidden = p.mlp.fc1.weight.shape[0]
    h = linear_fwd(xn2, p.mlp.fc1.weight, p.mlp.fc1.bias, M, C)
    dw = p.mlp.dwconv.dwconv
    act = ops.dwconv_fwd(h, rt.wdw(dw.weight), dw.bias, B, H, W, hidden, 1, 'gelu')
    Cout = p.mlp.fc2.weight.shape[0]
    x2 = linear_fwd(act, p.mlp.fc2.weight, p.mlp.fc2.bias, M, hidden, res=x1 if Cout == C else None, rowscale=dp2,
                    rows_per_scale=N, out_dtype=sd if Cout == C else None)
    saved = (x, m1, r1, xn, q, xs_pre, ms, rs, xs, kv, P, o, x1, m2, r2, xn2, h, act, Nk, dp1, dp2) if save else None
    return x2, saved


def mlp_bwd(dy, mlp, xin, h, act, B, H, W, Cin, dps=None, dy_scaled=None):
    """Backward of Mlp (fc1 -> dwconv3x3 -> GELU -> fc2); returns d(xin).  dy_scaled: dy * dps already computed by the producer
    of dy (the LayerNorm backward's second output)."""
    M = B * H * W
    hidden = mlp.fc1.weight.shape[0]
    dys = dy if dps is None else (dy_scaled if dy_scaled is not None else ops.sample_scale(dy, dps, B, dy.shape[1]))
    da = linear_bwd(dys, act, mlp.fc2.weight, mlp.fc2.bias, M, hidden)
    dw = mlp.dwconv.dwconv
    w9 = rt.wdw(dw.weight)
    dz = ops.dwconv_gelu_bwd_fused(h, w9, dw.bias, da, rt.grad(dw.weight).view(hidden, 9), rt.grad(dw.bias), B, H, W, hidden, 1)
    dh = ops.dwconv_bwd_data(dz, w9, B, H, W, hidden, 1, out=da)
    return linear_bwd(dh, xin, mlp.fc1.weight, mlp.fc1.bias, M, Cin)


def block_bwd(dy, p, saved, B, H, W, C, heads, sr, *, eps=1e-6, dy_scaled=None, next_scale=None):
    """dy_scaled: dy * dp2 (this block's MLP-branch DropPath factor) when the producer of dy already wrote it; next_scale: the
    per-sample factor the CONSUMER of this block's input gradient wants applied -> returns (dx, dx * next_scale)."""
    (x, m1, r1, xn, q, xs_pre, ms, rs, xs, kv, P, o, x1, m2, r2, xn2, h, act, Nk, dp1, dp2) = saved
    N = H * W
    M = B * N
    a = p.attn
    hd = C // heads
    dxn2 = mlp_bwd(dy, p.mlp, xn2, h, act, B, H, W, C, dp2, dy_scaled)
    if dp1 is None:
        dps = dx1 = ops.layernorm_bwd(dxn2, x1, p.norm2.weight, m2, r2, rt.grad(p.norm2.weight), rt.grad(p.norm2.bias), dres=dy)
    else:   # the attention branch's DropPath factor rides along as the LayerNorm backward's second output
        dx1, dps = ops.layernorm_bwd(dxn2, x1, p.norm2.weight, m2, r2, rt.grad(p.norm2.weight), rt.grad(p.norm2.bias), dres=dy,
                                     out_scale=dp1, rows_per_scale=N)
    do = linear_bwd(dps, o, a.proj.weight, a.proj.bias, M, C)
    dq, dkv = attention_bwd(do, q, kv, P, B, N, Nk, heads, C, hd ** -0.5)
    dxs = linear_bwd(dkv, xs, a.kv.weight, a.kv.bias, B * Nk, C)
    if sr > 1:
        dxs_pre = ops.layernorm_bwd(dxs, xs_pre, a.norm.weight, ms, rs, rt.grad(a.norm.weight), rt.grad(a.norm.bias))
        dxn = conv_bwd(dxs_pre, xn, a.sr.weight, a.sr.bias, B, H, W, sr, 0)
    else:
        dxn = dxs
    linear_bwd(dq, xn, a.q.weight, a.q.bias, M, C, dx_out=dxn, dx_beta=1.0)
    return ops.layernorm_bwd(dxn, x, p.norm1.weight, m1, r1, rt.grad(p.norm1.weight), rt.grad(p.norm1.bias), dres=dx1,
                             out_scale=next_scale, rows_per_scale=N)
