"""Thin tensor-level wrappers over the C ABI (include/cmda_hip.h).  No autograd here: the
differentiable building blocks live in cmda_amd/functional.py and call these for both passes."""
import ctypes
import os

import torch

from . import _lib as L
from ._lib import BF16, F32, GemmParams, View, c_f32, c_i32, c_i64, call, check_dev, dtype_tag, ptr, stream_of

_ESIZE = {torch.float32: 4, torch.bfloat16: 2}


def _chunk(t):
    return 16 // _ESIZE[t.dtype]


def plain_view(t, rows, cols, ld=None, batch_stride=0, offset=0, batch2_stride=0):
    """View of a row-major matrix living inside tensor `t` (element `offset` from its start)."""
    ld = cols if ld is None else ld
    es = _ESIZE[t.dtype]
    ch = 16 // es
    base = t.data_ptr() + offset * es
    vec_ok = int(base % 16 == 0 and ld % ch == 0 and batch_stride % ch == 0 and batch2_stride % ch == 0)
    v = View(ptr=base, ld=ld, R=rows, Cc=cols, batch_stride=batch_stride, batch2_stride=batch2_stride, conv=0,
             vec_ok=vec_ok,
             H=0, W=0, C=1, OH=1, OW=1, KH=1, KW=1, stride=1, pad=0, dil=1, in_dil=1, reflect=0)
    v._t, v._off = t, offset    # (the tensor behind the view: `_gemm_x3_big` re-points the view at its bf16 hi / lo copies)
    return v


def conv_view(x, B, H, W, C, KH, KW, stride, pad, dil=1, OH=None, OW=None, in_dil=1, reflect=0):
    """im2col view of NHWC tensor x[B,H,W,C]: r=(b,oh,ow), c=(kh,kw,ci)."""
    if OH is None:
        OH = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
        OW = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
    ch = _chunk(x)
    vec_ok = int(x.data_ptr() % 16 == 0 and C % ch == 0)
    # non-overlapping patches (the spatial-reduction convolutions): the same matrix, but its rows / K segments are contiguous
    # runs the kernels can fill like a plain operand (conv = 2)
    patch = (KH == KW == stride and pad == 0 and dil == 1 and in_dil == 1 and not reflect and H == OH * stride and W == OW * stride)
    v = View(ptr=x.data_ptr(), ld=0, R=B * OH * OW, Cc=KH * KW * C, batch_stride=0, batch2_stride=0, conv=2 if patch else 1, H=H, W=W, C=C,
             OH=OH, OW=OW, KH=KH, KW=KW, stride=stride, pad=pad, dil=dil, in_dil=in_dil, reflect=reflect,
             vec_ok=vec_ok)
    v._t, v._off = x, 0
    return v


ACT = {None: 0, 'none': 0, 'relu': 1, 'gelu': 2, 'tanh': 3}

# bench.py's roofline leg: when a list is installed here every GEMM launch is bracketed by events on the launch stream
GEMM_PROFILE = None
# which part of the path is issuing work (set by the modules through `site(...)`): bench.py splits the MFMA roofline by it --
# 'mit' (the MiT encoders: the blocks the 0.60 target is defined on), 'fusion', 'head', 'generator', 'other'
GEMM_SITE = 'other'


class site:
    """with ops.site('mit'): ...  -- tags every GEMM / fused-attention launch (and every queued weight gradient) issued inside"""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        global GEMM_SITE
        self.prev, GEMM_SITE = GEMM_SITE, self.name

    def __exit__(self, *exc):
        global GEMM_SITE
        GEMM_SITE = self.prev


def sited(name):
    """decorator form of `site`"""
    def deco(fn):
        import functools

        @functools.wraps(fn)
        def wrapped(*a, **k):
            with site(name):
                return fn(*a, **k)
        return wrapped
    return deco


def _attn_profile(flops, fn):
    """bench.py's roofline leg: bracket a fused-attention launch like a GEMM launch (MFMA work of the MiT blocks)"""
    if GEMM_PROFILE is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    GEMM_PROFILE.append((flops, e0, e1, 0, ('attention', GEMM_SITE)))
    return r
# cmda_gemm_params_t.tile_hint for every GEMM issued from here (0 = the library's heuristics); set by tuning sweeps / tests
GEMM_TILE_HINT = int(os.environ.get('CMDA_GEMM_TILE_HINT', '0'))   # cmda_gemm_params_t.tile_hint of every launch (tuning sweeps, forced-tile tests)


# ---- split-bf16 mode, LARGE problems: three launches of the bf16 kernels over operands split once in HBM -------------------------
# The register-staged split kernel (csrc/gemm_x3.hip) runs the decode head's 3 x 3 bottleneck at ~170 TFLOP/s (7.3 ms per call at
# 16 images) where the bf16 LDS-DMA kernel does 1.1 PFLOP/s; x = hi + lo with both halves stored as bf16 tensors of x's own layout
# turns the contraction into a_lo b_hi + a_hi b_lo + a_hi b_hi on that kernel, accumulated in the fp32 output (beta = 1 / atomics):
# same ~16 mantissa bits per product, 8 bytes of extra traffic per operand element.
X3_BIG_FLOPS = float(os.environ.get('CMDA_X3_BIG_GFLOP', '15')) * 1e9   # (bench, ms per step at 40 / 20 / 10 / 5 / 2 GFLOP: 149.2 / 144.7 / 145.1 / 150.7 / 162.3)


def split_bf16(t):
    """(hi, lo) bf16 tensors of t's shape: hi = bf16(t), lo = bf16(t - hi).  Nothing is cached: a captured iteration replays the
    split launch with that iteration's operand values (the weights of these problems are a few MB)."""
    check_dev(t)
    hi = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
    lo = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
    call('cmda_split_bf16', ptr(t), ptr(hi), ptr(lo), c_i64(t.numel()), stream_of(t))
    return hi, lo


X3_BIG_INTENSITY = float(os.environ.get('CMDA_X3_BIG_INTENSITY', 100))   # FLOP per byte moved by the split / accumulate passes


def _x3_big_ok(A, B, out, M, N, K, nb, act, rowscale, hold, defer, splits, c_patch, c_perm):
    if hold or act is not None or rowscale is not None or c_patch is not None or GEMM_TILE_HINT != 0:
        return False
    if 2.0 * M * N * K * nb < X3_BIG_FLOPS or out.dtype != torch.float32:
        return False
    moved = 16.0 * M * N * nb   # two more read + write passes over the fp32 output (beta = 1 launches)
    for v in (A, B):
        t = getattr(v, '_t', None)
        if t is None or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() % 8 or t.data_ptr() % 16:
            return False
        if v.conv:
            if v.C % 8:
                return False
        elif v.ld % 8 or v._off % 8 or v.batch_stride % 8 or v.batch2_stride % 8:
            return False
        moved += 12.0 * t.numel()   # the split pass: 4 bytes read, 2 x 2 written, the halves read again
    # the three launches pay for themselves where the contraction outweighs those passes (profiles/r05_x3_gemm.txt: 4096^3 at 204 FLOP
    # per moved byte 460 against 728 us on the register-staged split kernel; the head's pointwise convolution 262144 x 256 x 1024 at 32:
    # 1438 against 1119)
    return 2.0 * M * N * K * nb >= X3_BIG_INTENSITY * moved


def _x3_half_views(v):
    hi, lo = split_bf16(v._t)
    out = []
    for h in (hi, lo):
        w = View.from_buffer_copy(bytes(v))
        w.ptr = h.data_ptr() + v._off * 2
        w.vec_ok = 1
        w._t, w._off = h, v._off
        out.append(w)
    return out[0], out[1], (hi, lo)


def _gemm_x3_big(A, B, out, M, N, K, kw, colstats=None):
    """out = a_lo b_hi + a_hi b_lo + a_hi b_hi (bias / residual / caller's beta in the first launch; fused column statistics in the
    last one, whose epilogue stores the final values)"""
    a_hi, a_lo, ka = _x3_half_views(A)
    b_hi, b_lo, kb = _x3_half_views(B)
    keep = tuple(kw.pop('keep', ())) + ka + kb
    first = dict(kw)
    rest = dict(kw, bias=None, res=None)
    atomic = kw.get('atomic', False)
    if not atomic:
        rest['beta'] = 1.0
    colsum = kw.get('colsum')
    # the bias gradient (column sums of A = dy): sum of the two halves' column sums, taken in the launches that read them first
    gemm(a_lo, b_hi, out, M, N, K, **dict(first, dtype=1, colsum=colsum, keep=keep))
    gemm(a_hi, b_lo, out, M, N, K, **dict(rest, dtype=1, colsum=colsum, keep=keep))
    gemm(a_hi, b_hi, out, M, N, K, **dict(rest, dtype=1, colsum=None, keep=keep, colstats=colstats))
    return out


def gemm(A, B, out, M, N, K, *, a_kstrided=False, b_kstrided=False, ldc=None, batch=1, c_batch_stride=0,
         batch2=1, c_batch2_stride=0, res_batch2_stride=0, splits=1, alpha=1.0, beta=0.0, bias=None, act=None, res=None, ldres=None, res_batch_stride=0,
         rowscale=None, rows_per_scale=1, atomic=False, dtype=None, c_offset=0, colsum=None, c_patch=None, c_perm=None, defer=False, keep=(),
         hold=False, colstats=None):
    """out[m,n] = epi(alpha * sum_k A(m,k) B(n,k)); A/B are `View`s built by plain_view / conv_view.
    defer=True (weight gradients: nothing reads `out` before the pass ends): inside a deferral scope (`ln_deferral`) the launch is
    only QUEUED and goes out with the next `gemm_flush_deferred()` as part of a grouped launch; `keep` = the tensors behind the
    operand views (kept alive until then).
    hold=True: build the problem but do NOT launch it -- returns (params, meta, out, keep) for tools that time or inspect it.
    colstats=(ws, rows_per_group): column sums / sums of squares of the stored output accumulated into the BatchNorm workspace `ws`
    (`bn_stats_ws`: zero on entry) by the epilogue -- the statistics pass of the BatchNorm / InstanceNorm behind this convolution
    (`colstats_ok` says whether a problem qualifies)."""
    check_dev(out, bias, res, rowscale)
    if dtype == 2 and _x3_big_ok(A, B, out, M, N, K, batch * batch2, act, rowscale, hold, defer, splits, c_patch, c_perm):
        return _gemm_x3_big(A, B, out, M, N, K, dict(a_kstrided=a_kstrided, b_kstrided=b_kstrided, ldc=ldc, batch=batch,
                                                     c_batch_stride=c_batch_stride, batch2=batch2, c_batch2_stride=c_batch2_stride,
                                                     res_batch2_stride=res_batch2_stride, splits=splits, alpha=alpha, beta=beta, bias=bias,
                                                     res=res, ldres=ldres, res_batch_stride=res_batch_stride, atomic=atomic,
                                                     c_offset=c_offset, colsum=colsum, c_perm=c_perm, defer=defer, keep=keep), colstats=colstats)
    out_f32 = out.dtype == torch.float32
    p = GemmParams()
    p.A, p.B = A, B
    p.a_kstrided, p.b_kstrided = int(a_kstrided), int(b_kstrided)
    p.C = out.data_ptr() + c_offset * _ESIZE[out.dtype]
    p.ldc = N if ldc is None else ldc
    p.c_batch_stride, p.c_batch2_stride = c_batch_stride, c_batch2_stride
    p.M, p.N, p.K, p.batch, p.batch2, p.splits = M, N, K, batch, batch2, splits
    p.alpha, p.beta = alpha, beta
    p.bias = bias.data_ptr() if bias is not None else None
    p.act = ACT[act]
    p.res = res.data_ptr() if res is not None else None
    p.res_f32 = int(res is not None and res.dtype == torch.float32 and dtype == 1)   # fp32 residual stream of the bf16 mode
    p.ldres = (N if ldres is None else ldres)
    p.res_batch_stride, p.res_batch2_stride = res_batch_stride, res_batch2_stride
    p.rowscale = rowscale.data_ptr() if rowscale is not None else None
    p.rows_per_scale = rows_per_scale
    p.dtype = dtype
    p.out_f32 = int(out_f32)
    assert out_f32 or not atomic
    p.atomic = int(atomic)
    ok = (p.ldc % 4 == 0 and c_batch_stride % 4 == 0 and c_batch2_stride % 4 == 0 and p.C % 16 == 0)
    if res is not None:
        ok = ok and p.ldres % 4 == 0 and res_batch_stride % 4 == 0 and res_batch2_stride % 4 == 0 and res.data_ptr() % 16 == 0
    if bias is not None:
        ok = ok and bias.data_ptr() % 16 == 0
    p.c_vec_ok = int(ok)
    p.colsum = colsum.data_ptr() if colsum is not None else None
    if colstats is not None:
        check_dev(colstats[0])
        assert colstats[0].dtype == torch.float32 and not defer and not atomic
        p.colstats, p.colstats_rows = colstats[0].data_ptr(), colstats[1]
    p.tile_hint = GEMM_TILE_HINT
    if c_perm is not None:   # (Ci, KH*KW): atomic store of a conv weight gradient in the parameter's [Co,Ci,KH,KW] layout
        assert atomic and batch == 1 and batch2 == 1
        p.c_perm_ci, p.c_perm_cells = c_perm
    if c_patch is not None:  # (OW, KH, KW*Ci): store rows (b,oh,ow) x cols (kh,kw,ci) un-patchified into NHWC
        assert res is None and batch == 1 and batch2 == 1 and not atomic
        p.c_patch_ow, p.c_patch_kh, p.c_patch_kwci = c_patch
        p.c_vec_ok = int(p.C % 16 == 0 and c_patch[2] % 4 == 0 and (bias is None or bias.data_ptr() % 16 == 0))
    if defer:
        # deferred problems are launched AFTER the grouped ones of their flush whatever their list position (cmda_gemm_grouped): only
        # commutative accumulation may be deferred
        assert atomic and beta == 0.0, 'ops.gemm(defer=True) is for atomic accumulation only'
    if defer and _LN_DEFER['depth'] > 0 and GEMM_DEFER:
        es = 2 if dtype == 1 else 4
        nb = batch * batch2
        _GD['queues'].setdefault(GD_QUEUE_KEY or LN_LANE, []).append((p, (out, colsum) + tuple(keep), 2.0 * M * N * K * nb,
                                                      (M * K + N * K) * nb * es + 2 * M * N * nb * 4, GEMM_SITE))
        return out
    if hold or (GEMM_PROFILE is not None and out.is_cuda):
        es = 2 if dtype == 1 else 4
        nb = batch * batch2

        def _unique(v):  # bytes of the tensor behind an operand view (an im2col view re-reads, the tensor is counted once)
            return (v.R // max(1, v.OH * v.OW) * v.H * v.W * v.C if v.conv else v.R * v.Cc * nb) * es
        cbytes = M * N * nb * (4 if out_f32 else es) * (2 if (atomic or beta != 0.0) else 1)
        meta = (2.0 * M * N * K * nb, _unique(A) + _unique(B) + cbytes + (M * N * nb * es if res is not None else 0),
                (M, N, K, nb, splits, bool(A.conv or B.conv), bool(a_kstrided), bool(b_kstrided), bool(atomic), out_f32))
        if hold:
            return (p, meta, out, (bias, res, rowscale, colsum) + tuple(keep))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call('cmda_gemm', ctypes.byref(p), stream_of(out))
        e1.record()
        GEMM_PROFILE.append((meta[0], e0, e1, meta[1], meta[2] + (GEMM_SITE,)))
        return out
    call('cmda_gemm', ctypes.byref(p), stream_of(out))
    return out


# ---- deferred weight gradients: queued by gemm(defer=True) inside a deferral scope, launched in groups (cmda_gemm_grouped) ----------
import os as _os
# queue key override for gemm(defer=True) (None: the current lane).  The decode head's weight gradients are queued under their own
# key when they are to run in the TAIL of the backward pass (segmentors.train_bwd): the per-stage flushes of the encoders' backward
# passes, which run on the same lane in between, must not launch them.
GD_QUEUE_KEY = None
# work postponed to the same tail (closures: the head's depthwise weight-gradient kernels), run by `run_tail()`
_TAIL_FNS = []
GEMM_DEFER = _os.environ.get('CMDA_GEMM_DEFER', '1') != '0'    # False: defer=True launches in place (A/B switch for tuning, tests of the single-launch path)
_GD = {'queues': {}, 'plans': {}, 'arena': None, 'pinned_plans': False}
_GD_ARENA_BYTES = 192 << 20
_GD_EAGER_ARENA_BYTES = 64 << 20


def _gd_arena(dev, nbytes):
    """slice of the pinned host arena the grouped launches' tables are built in (a plain CPU tensor under the emulator).
    Two arenas: the MAIN one holds every plan made before or inside a capture -- a captured graph re-runs the upload kernel of its
    plans on every replay, so its slices are never reused; once a capture has pinned plans, EAGER plans made afterwards (backward
    passes whose operand pointers differ from the captured ones: allocator churn, another batch shape's warm-up) live in a second,
    RECYCLABLE arena that is rewound (one sync, un-captured plans dropped) when it fills -- pinned host memory stays bounded at two
    arenas whatever the training run does (one retired 192 MB arena per exhaustion before).  `_GD['recycles']` counts
    the rewinds."""
    nbytes = (nbytes + 255) // 256 * 256
    capturing = dev.type == 'cuda' and torch.cuda.is_current_stream_capturing()
    pin = dev.type == 'cuda'
    if _GD['pinned_plans'] and not capturing:
        a = _GD.get('eager_arena')
        if a is None or a[0].numel() < nbytes:
            a = _GD['eager_arena'] = [torch.empty(max(_GD_EAGER_ARENA_BYTES, nbytes), dtype=torch.uint8, pin_memory=pin), 0]
        if a[1] + nbytes > a[0].numel():
            if pin:
                torch.cuda.synchronize(dev)   # every upload that read the eager arena has run
            _GD['plans'] = {k: v for k, v in _GD['plans'].items() if v[3]}
            a[1] = 0
            _GD['recycles'] = _GD.get('recycles', 0) + 1
        lo = a[1]
        a[1] = lo + nbytes
        return a[0][lo:lo + nbytes], True
    a = _GD['arena']
    if a is None:
        if capturing:
            raise RuntimeError('grouped-GEMM arena: run one eager backward pass before capturing')
        a = _GD['arena'] = [torch.empty(_GD_ARENA_BYTES, dtype=torch.uint8, pin_memory=pin), 0]
    if a[1] + nbytes > a[0].numel():
        if capturing or _GD['pinned_plans']:
            raise RuntimeError('grouped-GEMM arena exhausted by captured plans: raise ops._GD_ARENA_BYTES')
        if pin:
            torch.cuda.synchronize(dev)   # eager plans only: every upload that read the arena has run
        _GD['plans'].clear()
        a[1] = 0
    lo = a[1]
    a[1] = lo + nbytes
    return a[0][lo:lo + nbytes], False


def tail_defer(fn):
    """run fn() now -- or, while a tail queue is open (GD_QUEUE_KEY set inside a deferral scope), together with that queue"""
    if GD_QUEUE_KEY is not None and _LN_DEFER['depth'] > 0:
        _TAIL_FNS.append(fn)
    else:
        fn()


def run_tail(key):
    """launch everything postponed under `key`: the queued weight gradients (grouped launch) and the postponed closures"""
    fns = list(_TAIL_FNS)
    del _TAIL_FNS[:]
    for fn in fns:
        fn()
    gemm_flush_deferred(from_lane=key)


def gemm_deferred_tensors(lane=None):
    """the tensors behind the queued problems of `lane` (default: the current one) -- what a side lane that launches them must keep alive"""
    lane = LN_LANE if lane is None else lane
    return [t for e in _GD['queues'].get(lane, ()) for t in e[1] if t is not None]


def gemm_flush_deferred(all_lanes=False, from_lane=None):
    """launch what gemm(defer=True) queued since the last flush: the CURRENT concurrency lane's queue by default (another lane's
    producers may still be running on their stream), every queue with all_lanes (after the lanes were joined).  from_lane: launch
    THAT lane's queue from here (a side lane entered behind the producers: the decode head's weight gradients next to the encoders'
    backward pass, segmentors.train_bwd)."""
    if from_lane is not None:
        lanes = [k for k in _GD['queues'] if k == from_lane or k.startswith(from_lane + '/wgrad')]
    else:
        lanes = list(_GD['queues']) if all_lanes else [k for k in _GD['queues'] if k == LN_LANE or k.startswith(LN_LANE + '/wgrad')]
    for lane in lanes:
        q = _GD['queues'].pop(lane, None)
        if not q:
            continue
        n = len(q)
        arr = (GemmParams * n)(*[e[0] for e in q])
        out0 = q[0][1][0]
        dev = out0.device
        key = (str(dev), bytes(arr))
        plan = _GD['plans'].get(key)
        upload = 0
        capturing = dev.type == 'cuda' and torch.cuda.is_current_stream_capturing()
        if plan is not None and capturing and plan[4]:
            plan = None   # made eagerly in the recyclable arena: a captured graph needs a slice that stays -> rebuild it in the main one
        if plan is None:
            nbytes = int(L.lib().cmda_gemm_grouped_ws_bytes(arr, c_i32(n)))
            host, recyclable = _gd_arena(dev, max(nbytes, 16))
            plan = _GD['plans'][key] = [host, torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev), nbytes, False, recyclable]
            upload = 1
        if capturing:
            _GD['pinned_plans'] = True
            plan[3] = True    # a captured graph replays this plan's upload + launch: its arena slice and device table stay
        host, devbuf, nbytes = plan[:3]
        prof = GEMM_PROFILE is not None and out0.is_cuda
        if prof:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        call('cmda_gemm_grouped', arr, c_i32(n), ptr(host), ptr(devbuf), c_i64(nbytes), c_i32(upload), stream_of(out0))
        if prof:
            e1.record()
            split = {}
            for e in q:
                split[e[4]] = split.get(e[4], 0.0) + e[2]
            GEMM_PROFILE.append((sum(e[2] for e in q), e0, e1, sum(e[3] for e in q), ('grouped', n, 0, 0, 0, False, True, True, True, True, split)))


def layernorm_fwd(x, gamma, beta, eps, save_stats=True, out=None, out_dtype=None):
    """y = LayerNorm(x); y takes out's dtype, else out_dtype, else x's (x fp32 -> y bf16 and back: the fp32 residual stream)"""
    check_dev(x, gamma, beta, out)
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty(x.shape, dtype=out_dtype or x.dtype, device=x.device) if out is None else out
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    call('cmda_layernorm_fwd2', ptr(x), dtype_tag(x), ptr(gamma), ptr(beta), ptr(y), dtype_tag(y), ptr(mean), ptr(rstd), c_i64(rows),
         c_i32(C), c_f32(eps), stream_of(x))
    return y, mean, rstd


_LN_WS = {}
_LN_WS_MAX = {}
LN_LANE = 'main'   # set by runtime.lane: LayerNorm backward passes running side by side must not share one workspace


def _ln_ws(device, n):
    """persistent zero-initialised LayerNorm-backward workspace (the kernel pair leaves it zeroed; calls on one stream are
    ordered, so one buffer per device and concurrency lane is enough)"""
    dkey = (device.type, device.index)
    _LN_WS_MAX[dkey] = max(_LN_WS_MAX.get(dkey, 64 * 2 * 1024), n)
    key = dkey + (LN_LANE,)
    ws = _LN_WS.get(key)
    if ws is None or ws.numel() < n:
        ws = _LN_WS[key] = torch.zeros(_LN_WS_MAX[dkey], dtype=torch.float32, device=device)
    return ws


def ln_ws_prealloc(device, lanes):
    """create the workspaces of the given lanes NOW (eagerly), sized for the largest request seen so far -- so that a graph
    capture never allocates one from its private pool"""
    dkey = (device.type, device.index)
    n = _LN_WS_MAX.get(dkey, 64 * 2 * 1024)
    for ln in lanes:
        key = dkey + (ln,)
        if key not in _LN_WS or _LN_WS[key].numel() < n:
            _LN_WS[key] = torch.zeros(n, dtype=torch.float32, device=device)


# Deferred LayerNorm parameter gradients: inside `ln_deferral()` every layernorm_bwd leaves its per-block partial sums in a
# workspace owned by that layer (keyed by its dgamma buffer) and ONE cmda_layernorm_fold_batch launch at the end of the scope
# folds all of them into dgamma / dbeta -- the ~700 finalize launches per UDA step this replaces only fed the optimizer.
_LN_DEFER = {'depth': 0, 'regions': {}, 'touched': {}, 'plans': {}}


_ARENA = {}


def _host_table(raw, dev):
    """A small descriptor table the kernels read in place.  On the GPU it lives in PINNED HOST memory (a few KB per launch over the
    bus): a new set of layers may show up while a stream is capturing (the per-lane folds of a segmented capture differ from the
    eager iteration's), and neither a host-to-device copy nor an allocation is capturable -- a slice of an arena allocated up
    front is."""
    if dev.type != 'cuda':
        return raw
    arena = _ARENA.get('a')
    if arena is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('descriptor arena: run one eager backward pass before capturing')
        arena = _ARENA['a'] = [torch.empty(16 << 20, dtype=torch.uint8, pin_memory=True), 0]
    lo = arena[1]
    if lo + raw.numel() > arena[0].numel():
        raise RuntimeError('descriptor arena exhausted')
    tab = arena[0][lo:lo + raw.numel()]
    tab.copy_(raw)   # host-to-host
    arena[1] = lo + (raw.numel() + 63) // 64 * 64
    return tab


class ln_deferral:
    def __enter__(self):
        _LN_DEFER['depth'] += 1
        return self

    def __exit__(self, *exc):
        _LN_DEFER['depth'] -= 1
        if _LN_DEFER['depth'] == 0:
            if exc[0] is None:
                fns = list(_TAIL_FNS)   # (a tail nobody ran: run it here)
                del _TAIL_FNS[:]
                for fn in fns:
                    fn()
                ln_fold_deferred(all_lanes=True)
            else:   # the pass died half way: drop what it queued / touched instead of folding it into the next pass
                del _TAIL_FNS[:]
                _LN_DEFER['touched'] = {}
                _CG['touched'] = {}
                _GD['queues'].clear()
        return False


def _take_lanes(store, all_lanes):
    """pop the entries of the CURRENT lane and of the lanes forked below it that were joined back (`<lane>/wgrad...`: the
    batched weight-gradient closures of runtime.lane_batch register their work there) -- or of every lane"""
    if all_lanes:
        keys = list(store)
    else:
        keys = [k for k in store if k == LN_LANE or k.startswith(LN_LANE + '/wgrad')]
    out = {}
    for k in keys:
        out.update(store.pop(k))
    return out


def _ln_region(dgamma, dbeta, C):
    key = dgamma.data_ptr()
    r = _LN_DEFER['regions'].get(key)
    if r is None or r[0].device != dgamma.device:
        nslots = L.lib().cmda_layernorm_slots()
        r = _LN_DEFER['regions'][key] = (torch.zeros(nslots * 2 * C, dtype=torch.float32, device=dgamma.device), dgamma, dbeta, C, nslots)
    _LN_DEFER['touched'].setdefault(LN_LANE, {})[key] = r
    return r[0]


def ln_fold_deferred(all_lanes=False):
    """fold every workspace touched since the last fold into its dgamma / dbeta: one launch per device.  Only the CURRENT
    concurrency lane's workspaces by default (another lane's LayerNorm backward kernels may still be running on their stream);
    the end of the scope, which follows the lanes' joins, folds them all."""
    gemm_flush_deferred(all_lanes)   # queued weight gradients first: the convolution ones land in the shadows drained next
    conv_grad_drain(all_lanes)
    touched = _take_lanes(_LN_DEFER['touched'], all_lanes)
    if not touched:
        return
    import numpy as np
    by_dev = {}
    for key, r in touched.items():
        by_dev.setdefault(r[0].device, []).append((key, r))
    for dev, items in by_dev.items():
        items.sort(key=lambda kv: kv[0])
        pkey = (dev, tuple(k for k, _ in items))
        plan = _LN_DEFER['plans'].get(pkey)
        if plan is None:
            desc = np.zeros(len(items), dtype=[('ws', '<u8'), ('dg', '<u8'), ('db', '<u8'), ('C', '<i4'), ('n', '<i4')])
            for i, (_, (ws, dg, db, C, nslots)) in enumerate(items):
                desc[i] = (ws.data_ptr(), dg.data_ptr(), db.data_ptr(), C, nslots)
            tab = _host_table(torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()), dev)
            plan = _LN_DEFER['plans'][pkey] = (tab, len(items), max(r[3] for _, r in items), dev)
        call('cmda_layernorm_fold_batch', ptr(plan[0]), c_i32(plan[1]), c_i32(plan[2]), stream_of(items[0][1][0]))


# Deferred convolution weight gradients: inside the same scope a conv weight gradient is accumulated by the GEMM's atomics in the
# GEMM's own column order (kh, kw, ci) -- coalesced -- into a persistent zeroed fp32 shadow [Co,KH,KW,Ci] of the parameter's
# gradient, and ONE batched launch per fold point moves all shadows into the [Co,Ci,KH,KW] gradients and clears them (the
# permuted atomic store it replaces cost 20-84 us per spatial-reduction conv against 10-14 us, tools/dbg/srconv_dbg.py).
_CG = {'regions': {}, 'touched': {}, 'plans': {}}


def conv_grad_shadow(grad, ci_pad=0):
    """grad: fp32 [Co,Ci,KH,KW] parameter gradient -> its [Co, KH*KW*Ci] shadow, or None outside a deferral scope.  ci_pad > Ci:
    the shadow carries the padded channels of the GEMM ([Co, KH*KW*ci_pad]); the drain keeps the real ones."""
    Co, Ci, KH, KW = grad.shape
    if ci_pad <= Ci and not grad.is_contiguous() and grad.permute(0, 2, 3, 1).is_contiguous():
        # channels-last stored gradient (optim.FlatAdamW): its memory is the GEMM's own [Co][KH][KW][Ci] order -- accumulate in place,
        # nothing to drain
        return grad.permute(0, 2, 3, 1).reshape(Co, KH * KW * Ci)
    if _LN_DEFER['depth'] == 0:
        return None
    key = grad.data_ptr()
    r = _CG['regions'].get(key)
    cs = max(Ci, ci_pad)
    if r is None or r[0].device != grad.device or r[0].shape[1] != KH * KW * cs:
        r = _CG['regions'][key] = (torch.zeros(Co, KH * KW * cs, dtype=torch.float32, device=grad.device), grad)
    _CG['touched'].setdefault(LN_LANE, {})[key] = r
    return r[0]


def conv_grad_drain(all_lanes=False):
    touched = _take_lanes(_CG['touched'], all_lanes)
    if not touched:
        return
    import numpy as np
    by_dev = {}
    for key, r in touched.items():
        by_dev.setdefault(r[0].device, []).append((key, r))
    for dev, items in by_dev.items():
        items.sort(key=lambda kv: kv[0])
        pkey = (dev, tuple(k for k, _ in items))
        plan = _CG['plans'].get(pkey)
        if plan is None:
            desc = np.zeros(len(items), dtype=[('src', '<u8'), ('dst', '<u8'), ('d', '<i4', 4), ('p', '<i4', 4), ('flip', '<i4'),
                                               ('mode', '<i4'), ('total', '<i8')])
            blocks = []
            for i, (_, (sh, g)) in enumerate(items):
                Co, Ci, KH, KW = g.shape
                cs = sh.shape[1] // (KH * KW)   # channels of the shadow (> Ci: padded)
                desc[i] = (sh.data_ptr(), g.data_ptr(), (Co, KH, KW, cs), (0, 3, 1, 2), ((4 << 8) | (Ci << 16)) if cs != Ci else 0, 2, g.numel())
                blocks += [(i, b) for b in range((g.numel() + 1023) // 1024)]
            tab = _host_table(torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()), dev)
            blk = _host_table(torch.from_numpy(np.asarray(blocks, dtype=np.int32).reshape(-1).view(np.uint8).copy()), dev)
            plan = _CG['plans'][pkey] = (tab, blk, len(blocks))
        call('cmda_permute4_batch', ptr(plan[0]), ptr(plan[1]), c_i32(plan[2]), stream_of(items[0][1][0]))


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dres=None, out_scale=None, rows_per_scale=0):
    """returns dx, or (dx, dx * out_scale[row // rows_per_scale]) when a per-sample scale is given (DropPath of the consumer)"""
    check_dev(dy, x, gamma, mean, rstd, dgamma, dbeta, dres, out_scale)
    C = x.shape[-1]
    rows = x.numel() // C
    dx = torch.empty_like(dy)       # (x may be the fp32 residual stream while the gradients travel in the compute dtype)
    dxs = torch.empty_like(dy) if out_scale is not None else None
    if _LN_DEFER['depth'] > 0:
        ws, dg, db = _ln_region(dgamma, dbeta, C), None, None
    else:
        ws, dg, db = _ln_ws(x.device, L.lib().cmda_layernorm_bwd_ws_floats(rows, C)), dgamma, dbeta
    call('cmda_layernorm_bwd2', ptr(dy), ptr(x), dtype_tag(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dg),
         ptr(db), ptr(ws), c_i64(rows), c_i32(C), ptr(out_scale), c_i64(rows_per_scale), ptr(dxs), dtype_tag(dy), stream_of(x))
    return dx if out_scale is None else (dx, dxs)


def permute4(src, dst, dims, perm, flipmask=0, accumulate=False):
    """dst (contiguous, dims[perm]) = permute(src viewed as `dims`), with optional axis flips / accumulate."""
    check_dev(src, dst)
    d = list(dims) + [1] * (4 - len(dims))
    p = list(perm) + list(range(len(perm), 4))
    call('cmda_permute4', ptr(src), ptr(dst), *[c_i32(v) for v in d], *[c_i32(v) for v in p], c_i32(flipmask),
         c_i32(int(accumulate)), dtype_tag(src), dtype_tag(dst), stream_of(src))
    return dst


def conv_co1_ok(x, C, K, pad):
    return x.dtype in (torch.bfloat16, torch.float32) and C == 64 and K == 7 and pad == 3


def conv_co1(x, w, bias, B, H, W, C, K, pad, reflect, act):
    """x [B*H*W, C] NHWC, w [K*K*C] khwc (activation dtype) -> fp32 [B,H,W] = act(bias + conv): one output channel"""
    check_dev(x, w, bias)
    out = torch.empty(B, H, W, dtype=torch.float32, device=x.device)
    call('cmda_conv_co1', ptr(x), ptr(w), ptr(bias), ptr(out), c_i32(B), c_i32(H), c_i32(W), c_i32(C), c_i32(K), c_i32(pad),
         c_i32(int(reflect)), c_i32(ACT[act]), dtype_tag(x), stream_of(x))
    return out


def cast_pad_cols(src32, cp, dtype):
    """fp32 [rows, c] -> dtype [rows, cp], columns >= c zero"""
    check_dev(src32)
    rows, c = src32.shape
    dst = torch.empty(rows, cp, dtype=dtype, device=src32.device)
    call('cmda_cast_pad_cols', ptr(src32), ptr(dst), c_i64(rows), c_i32(c), c_i32(cp), dtype_tag(dst), stream_of(src32))
    return dst


def rows_fill(out, bias):
    """out fp32 [rows, C] = bias[C] broadcast (zeros when bias is None)"""
    check_dev(out, bias)
    call('cmda_rows_fill', ptr(out), ptr(bias), c_i64(out.shape[0]), c_i32(out.shape[1]), stream_of(out))
    return out


def nchw_to_nhwc_pad(src, dst, B, C, HW, cpad):
    """src fp32 NCHW [B,C,H,W] -> dst [B*HW, cpad] (activation dtype), channels >= C zero"""
    check_dev(src, dst)
    call('cmda_nchw_to_nhwc_pad', ptr(src), ptr(dst), c_i32(B), c_i32(C), c_i64(HW), c_i32(cpad), dtype_tag(dst), stream_of(src))
    return dst


def permute4_batch(desc, blocks, nblocks):
    """desc: DEVICE uint8 tensor holding an array of cmda_permute_desc_t; blocks: DEVICE int32 [nblocks, 2]"""
    check_dev(desc, blocks)
    call('cmda_permute4_batch', ptr(desc), ptr(blocks), c_i32(nblocks), stream_of(desc))


_ZERO_WS = {}


def zero_ws(device, n):
    """persistent fp32 accumulation workspace, ZERO on entry by contract: whoever accumulates into it drains it with cast_clear
    (one buffer per device and concurrency lane; calls on one stream are ordered)"""
    key = (device.type, device.index, LN_LANE)
    ws = _ZERO_WS.get(key)
    if ws is None or ws.numel() < n:
        ws = _ZERO_WS[key] = torch.zeros(max(n, 1 << 20), dtype=torch.float32, device=device)
    return ws[:n]


def zero_ws_prealloc(device, lanes, n=1 << 22):
    for ln in lanes:
        key = (device.type, device.index, ln)
        if key not in _ZERO_WS or _ZERO_WS[key].numel() < n:
            _ZERO_WS[key] = torch.zeros(n, dtype=torch.float32, device=device)


def cast_clear(src32, dtype):
    """returns src32 cast to `dtype` and zeroes src32 (one launch)"""
    check_dev(src32)
    dst = torch.empty(src32.shape, dtype=dtype, device=src32.device)
    call('cmda_cast_clear', ptr(src32), ptr(dst), c_i64(src32.numel()), dtype_tag(dst), stream_of(src32))
    return dst


def cast(src, dtype):
    if src.dtype == dtype:
        return src
    dst = torch.empty(src.shape, dtype=dtype, device=src.device)
    return permute4(src, dst, (src.numel(), 1, 1, 1), (0, 1, 2, 3))


def colsum(x, out, M, N, ld=None, offset=0):
    """out[N] += column sums of the [M,N] matrix at element `offset` of x with row pitch ld."""
    check_dev(x, out)
    call('cmda_colsum', L.c_vp(x.data_ptr() + offset * _ESIZE[x.dtype]), ptr(out), c_i64(M), c_i32(N),
         c_i64(N if ld is None else ld), dtype_tag(x), stream_of(x))
    return out


def axpby(x, y, a, b, out=None):
    check_dev(x, y)
    out = torch.empty_like(x) if out is None else out
    call('cmda_axpby', ptr(x), ptr(y), ptr(out), c_f32(a), c_f32(b), c_i64(x.numel()), dtype_tag(x), stream_of(x))
    return out


def ema_update(ema, param, alpha, mirror=None):
    """ema = alpha * ema + (1 - alpha) * param; mirror: optional bf16 tensor of the same length that receives the result too"""
    check_dev(ema, param, mirror)
    call('cmda_ema_update', ptr(ema), ptr(param), c_f32(alpha), c_i64(ema.numel()), ptr(mirror), stream_of(ema))


def adamw_step(p, g, m, v, lr, beta1, beta2, eps, wd, step, p_bf16=None):
    check_dev(p, g, m, v, p_bf16)
    call('cmda_adamw_step', ptr(p), ptr(g), ptr(m), ptr(v), ptr(p_bf16), c_i64(p.numel()), c_f32(lr), c_f32(beta1),
         c_f32(beta2), c_f32(eps), c_f32(wd), c_i32(step), stream_of(p))


def class_mix(src, tgt, src_label, classes, channels_last=False):
    """src/tgt [B,C,H,W] (or [B,H,W,C]); src_label int64 [B,H,W]; classes int64 [B,K] padded with -1."""
    check_dev(src, tgt, src_label, classes)
    out = torch.empty_like(src)
    B = src.shape[0]
    HW = src_label.shape[-2] * src_label.shape[-1]
    Cch = src.numel() // (B * HW)
    call('cmda_class_mix', ptr(src), ptr(tgt), ptr(out), ptr(src_label), ptr(classes), c_i32(classes.shape[1]), c_i32(B),
         c_i32(HW), c_i32(Cch), c_i32(int(channels_last)), dtype_tag(src), stream_of(src))
    return out


def class_mix_label(src, tgt, src_label, classes):
    check_dev(src, tgt, src_label, classes)
    out = torch.empty_like(src)
    B = src.shape[0]
    call('cmda_class_mix_label', ptr(src), ptr(tgt), ptr(out), ptr(src_label), ptr(classes), c_i32(classes.shape[1]),
         c_i32(B), c_i32(src.numel() // B), stream_of(src))
    return out


def softmax_fwd_(s, rows, L, alpha):
    check_dev(s)
    call('cmda_softmax_fwd', ptr(s), c_i64(rows), c_i32(L), c_f32(alpha), dtype_tag(s), stream_of(s))
    return s


def softmax_bwd_(p, dp, rows, L, alpha):
    check_dev(p, dp)
    call('cmda_softmax_bwd', ptr(p), ptr(dp), c_i64(rows), c_i32(L), c_f32(alpha), dtype_tag(p), stream_of(p))
    return dp


def attention_fused_ok(q, Nk, heads, C, need_grad=True, x3=False):
    """the fused kernels hold every key of a (batch, head) in LDS: up to 256 with a backward pass to follow, up to 320 forward-only
    (inference on 440 x 640 frames: 260 / 280 keys, encoder_decoder.py:897-936).  x3: the split-bf16 mode (fp32 storage, runtime.gemm_x3)
    has instances of its own -- K / V as hi + lo bf16 images in LDS, up to 256 keys"""
    if x3 and q.dtype == torch.float32:
        return C == heads * 64 and 0 < Nk <= 256 and not ATTN_X3_OFF
    return q.dtype == torch.bfloat16 and C == heads * 64 and 0 < Nk <= (256 if need_grad else 320)


ATTN_X3_OFF = os.environ.get('CMDA_ATTN_X3', '1') == '0'   # split-bf16 mode on the unfused GEMM + softmax path (same-box A/B)


def _kv_split(kv):
    """hi / lo bf16 halves of an fp32 kv tensor, split ONCE per attention call (cmda_split_bf16) and kept on the tensor for the
    backward pass, which reads the same kv"""
    pair = getattr(kv, '_cmda_split', None)
    if pair is None or pair[2] != kv._version:
        hi, lo = split_bf16(kv)
        pair = kv._cmda_split = (hi, lo, kv._version)
    return pair[0], pair[1]


def attention_fused_fwd(q, kv, B, N, Nk, heads, C, scale):
    check_dev(q, kv)
    o = torch.empty(B * N, C, dtype=q.dtype, device=q.device)
    if q.dtype == torch.float32:   # split-bf16 instances (fp32 storage)
        hi, lo = _kv_split(kv)
        _attn_profile(4.0 * B * N * Nk * C, lambda: call('cmda_attention_fwd_x3', ptr(q), ptr(hi), ptr(lo), ptr(o), c_i32(B), c_i32(N),
                                                         c_i32(Nk), c_i32(heads), c_i32(C), c_f32(scale), stream_of(q)))
        return o
    _attn_profile(4.0 * B * N * Nk * C, lambda: call('cmda_attention_fwd', ptr(q), ptr(kv), ptr(o), c_i32(B), c_i32(N), c_i32(Nk),
                                                     c_i32(heads), c_i32(C), c_f32(scale), dtype_tag(q), stream_of(q)))
    return o


def attention_bwd_direct(B, N, Nk, heads):
    """True: the fused backward stores dK | dV straight as bf16 (few queries: one block per key slice walks them all)"""
    return bool(L.lib().cmda_attention_bwd_direct(int(B), int(N), int(Nk), int(heads)))


def attention_fused_bwd(q, kv, do, dkv32, B, N, Nk, heads, C, scale, dkv16=None):
    """returns dq; dK | dV accumulate into dkv32 (fp32 [B*Nk, 2C], zero on entry) or -- direct mode -- are stored into dkv16 (bf16)"""
    check_dev(q, kv, do, dkv32, dkv16)
    dq = torch.empty(B * N, C, dtype=q.dtype, device=q.device)
    stats = torch.empty(B * N * heads * 2, dtype=torch.float32, device=q.device)
    if q.dtype == torch.float32:   # split-bf16 instances: `dkv16` is then the fp32 dK | dV tensor of the direct mode
        hi, lo = _kv_split(kv)
        _attn_profile(10.0 * B * N * Nk * C, lambda: call('cmda_attention_bwd_x3', ptr(q), ptr(hi), ptr(lo), ptr(do), ptr(dq), ptr(dkv32),
                                                          ptr(dkv16), ptr(stats), c_i32(B), c_i32(N), c_i32(Nk), c_i32(heads), c_i32(C),
                                                          c_f32(scale), stream_of(q)))
        return dq
    _attn_profile(10.0 * B * N * Nk * C, lambda: call('cmda_attention_bwd', ptr(q), ptr(kv), ptr(do), ptr(dq), ptr(dkv32), ptr(dkv16),
                                                      ptr(stats), c_i32(B), c_i32(N), c_i32(Nk), c_i32(heads), c_i32(C), c_f32(scale),
                                                      dtype_tag(q), stream_of(q)))
    return dq


def dwconv_fwd(x, w, bias, B, H, W, C, dil=1, act=None, colstats=None):
    """colstats=(ws, images_per_group): the BatchNorm statistics of y on the way (dilated walk, no activation: cmda_dwconv3x3_fwd_stats)"""
    check_dev(x, w, bias)
    y = torch.empty_like(x)
    if colstats is not None:
        assert act is None and dil >= 2
        check_dev(colstats[0])
        call('cmda_dwconv3x3_fwd_stats', ptr(x), ptr(w), ptr(bias), ptr(y), c_i32(B), c_i32(H), c_i32(W), c_i32(C), c_i32(dil),
             ptr(colstats[0]), c_i32(colstats[1]), dtype_tag(x), stream_of(x))
        return y
    call('cmda_dwconv3x3_fwd', ptr(x), ptr(w), ptr(bias), ptr(y), c_i32(B), c_i32(H), c_i32(W), c_i32(C), c_i32(dil),
         c_i32(ACT[act]), dtype_tag(x), stream_of(x))
    return y


def dwconv_gelu_bwd_prep(x, w, bias, da, B, H, W, C, dil=1):
    check_dev(x, w, bias, da)
    dz = torch.empty_like(x)
    call('cmda_dwconv3x3_gelu_bwd_prep', ptr(x), ptr(w), ptr(bias), ptr(da), ptr(dz), c_i32(B), c_i32(H), c_i32(W),
         c_i32(C), c_i32(dil), dtype_tag(x), stream_of(x))
    return dz


def dwconv_gelu_bwd_fused(x, w, bias, da, dw, dbias, B, H, W, C, dil=1):
    """dz = da * gelu'(conv(x) + bias) and the depthwise weight / bias gradients (accumulated) in one pass; returns dz"""
    check_dev(x, w, bias, da, dw, dbias)
    dz = torch.empty_like(x)
    call('cmda_dwconv3x3_gelu_bwd_fused', ptr(x), ptr(w), ptr(bias), ptr(da), ptr(dz), ptr(dw), ptr(dbias), c_i32(B), c_i32(H),
         c_i32(W), c_i32(C), c_i32(dil), dtype_tag(x), stream_of(x))
    return dz


def dwconv_bwd_data(dy, w, B, H, W, C, dil=1, out=None, accumulate=False):
    check_dev(dy, w, out)
    dx = torch.empty_like(dy) if out is None else out
    call('cmda_dwconv3x3_bwd_data', ptr(dy), ptr(w), ptr(dx), c_i32(B), c_i32(H), c_i32(W), c_i32(C), c_i32(dil),
         c_i32(int(accumulate)), dtype_tag(dy), stream_of(dy))
    return dx


def dwconv_bwd_weight(dz, x, dw, dbias, B, H, W, C, dil=1):
    check_dev(dz, x, dw, dbias)
    call('cmda_dwconv3x3_bwd_weight', ptr(dz), ptr(x), ptr(dw), ptr(dbias), c_i32(B), c_i32(H), c_i32(W), c_i32(C),
         c_i32(dil), dtype_tag(x), stream_of(x))


def bilinear_fwd(x, y, B, IH, IW, OH, OW, C, ldy=None, coff=0):
    check_dev(x, y)
    call('cmda_bilinear_fwd', ptr(x), ptr(y), c_i32(B), c_i32(IH), c_i32(IW), c_i32(OH), c_i32(OW), c_i32(C),
         c_i32(C if ldy is None else ldy), c_i32(coff), dtype_tag(x), stream_of(x))
    return y


def bilinear_bwd(dy, dx, B, IH, IW, OH, OW, C, ldy=None, coff=0):
    check_dev(dy, dx)
    call('cmda_bilinear_bwd', ptr(dy), ptr(dx), c_i32(B), c_i32(IH), c_i32(IW), c_i32(OH), c_i32(OW), c_i32(C),
         c_i32(C if ldy is None else ldy), c_i32(coff), dtype_tag(dy), stream_of(dy))
    return dx


BN_FUSED_STATS = os.environ.get('CMDA_BN_FUSED_STATS', '1') != '0'   # statistics of conv -> BN / IN pairs in the GEMM epilogue (A/B switch)
_BN_WS = {}


def bn_stats_ws(device, groups, C):
    """the persistent ZERO workspace a GEMM epilogue accumulates BatchNorm statistics into (gemm(colstats=...)); bn_train_fwd /
    bn_train_fwd2 with stats_ws hand it back zeroed, so one buffer per device and concurrency lane serves every layer (calls on one
    stream are ordered).  None where the fused statistics are switched off."""
    if not BN_FUSED_STATS:
        return None
    n = 8 * int(L.lib().cmda_bn_ws_floats(2048))
    need = groups * int(L.lib().cmda_bn_ws_floats(C))
    key = (device.type, device.index, LN_LANE)
    ws = _BN_WS.get(key)
    if ws is None or ws.numel() < need:
        if device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
            return None   # (a lane nobody pre-allocated for, met inside a capture: the separate statistics pass)
        ws = _BN_WS[key] = torch.zeros(max(n, need), dtype=torch.float32, device=device)
    return ws[:need]


def bn_ws_prealloc(device, lanes):
    for ln in lanes:
        key = (device.type, device.index, ln)
        if key not in _BN_WS:
            _BN_WS[key] = torch.zeros(8 * int(L.lib().cmda_bn_ws_floats(2048)), dtype=torch.float32, device=device)


def colstats_ok(rows_per_group, N):
    """may the convolution in front of a BatchNorm / InstanceNorm over groups of `rows_per_group` rows and N channels take the
    statistics in its epilogue (cmda_gemm_params_t.colstats: whole 256-row tiles per group, 16-byte column quads)"""
    return BN_FUSED_STATS and rows_per_group > 0 and rows_per_group % 256 == 0 and N % 4 == 0


def bn_train_fwd(x, gamma, beta, y, running_mean, running_var, M, C, eps, momentum, relu, ldy=None, coff=0, groups=1, order=None,
                 stats_ws=None):
    """M = rows PER GROUP; x / y hold `groups` consecutive blocks of M rows (own statistics each); returns mean, rstd [groups, C].
    stats_ws: the workspace the producing GEMM's epilogue filled (gemm(colstats=(ws, M))): no statistics pass over x"""
    check_dev(x, gamma, beta, y, running_mean, running_var, stats_ws)
    mean = torch.empty(groups, C, dtype=torch.float32, device=x.device)
    rstd = torch.empty(groups, C, dtype=torch.float32, device=x.device)
    ws = stats_ws if stats_ws is not None else torch.empty(groups * L.lib().cmda_bn_ws_floats(C), dtype=torch.float32, device=x.device)
    order_c = (ctypes.c_int * groups)(*order) if order is not None else None
    call('cmda_bn_train_fwd', ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), ptr(running_mean),
         ptr(running_var), ptr(ws), c_i64(M), c_i32(C), c_f32(eps), c_f32(momentum), c_i32(int(relu)),
         c_i32(C if ldy is None else ldy), c_i32(coff), c_i32(groups), order_c, c_i32(int(stats_ws is not None)), dtype_tag(x), stream_of(x))
    return mean, rstd


def bn_train_fwd2(x, gamma, beta, y, M, C, eps, relu, groups=1, res32=None, y2=None, ldy=None, coff=0, stats_ws=None):
    """cmda_bn_train_fwd2: x and y of independent storage types, no running statistics; res32 (fp32 [groups*M, C]) is added after
    the normalisation, y2 (bf16 [groups*M, C]) receives a copy of the result.  Returns mean, rstd [groups, C].  stats_ws: bn_train_fwd"""
    check_dev(x, gamma, beta, y, res32, y2, stats_ws)
    assert res32 is None or res32.dtype == torch.float32
    assert y2 is None or y2.dtype == torch.bfloat16
    mean = torch.empty(groups, C, dtype=torch.float32, device=x.device)
    rstd = torch.empty(groups, C, dtype=torch.float32, device=x.device)
    ws = stats_ws if stats_ws is not None else torch.empty(groups * L.lib().cmda_bn_ws_floats(C), dtype=torch.float32, device=x.device)
    call('cmda_bn_train_fwd2', ptr(x), dtype_tag(x), ptr(gamma), ptr(beta), ptr(y), dtype_tag(y), ptr(mean), ptr(rstd), None, None,
         ptr(ws), c_i64(M), c_i32(C), c_f32(eps), c_f32(0.0), c_i32(int(relu)), c_i32(C if ldy is None else ldy), c_i32(coff),
         c_i32(groups), None, ptr(res32), ptr(y2), c_i32(int(stats_ws is not None)), stream_of(x))
    return mean, rstd


def bn_apply(x, mean, rstd, gamma, beta, y, M, C, relu, ldy=None, coff=0):
    check_dev(x, mean, rstd, gamma, beta, y)
    call('cmda_bn_apply', ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(y), c_i64(M), c_i32(C),
         c_i32(int(relu)), c_i32(C if ldy is None else ldy), c_i32(coff), dtype_tag(x), stream_of(x))


def bn_train_bwd(dy, x, mean, rstd, gamma, beta, dgamma, dbeta, M, C, relu, lddy=None, coff=0, groups=1):
    check_dev(dy, x, mean, rstd, gamma, beta, dgamma, dbeta)
    dx = torch.empty_like(x)
    ws = torch.empty(groups * L.lib().cmda_bn_ws_floats(C), dtype=torch.float32, device=x.device)
    call('cmda_bn_train_bwd', ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(dx), ptr(dgamma),
         ptr(dbeta), ptr(ws), c_i64(M), c_i32(C), c_i32(int(relu)), c_i32(C if lddy is None else lddy), c_i32(coff),
         c_i32(groups), dtype_tag(x), stream_of(x))
    return dx


def ce_upsample_fwd(logits, label, weight, H, W, ignore_index=255, acc=None):
    """logits fp32 NHWC [B,h,w,nc]; returns (acc[2] = (sum w*nll, #correct), lse[B,H,W]).  acc: optional ZEROED fp32 [2] to
    accumulate into (a slice of one buffer shared by the loss terms of a pass)"""
    check_dev(logits, label, weight, acc)
    B, h, w, nc = logits.shape
    lse = torch.empty(B, H, W, dtype=torch.float32, device=logits.device)
    if acc is None:
        acc = torch.zeros(2, dtype=torch.float32, device=logits.device)
    call('cmda_ce_upsample_fwd', ptr(logits), ptr(label), ptr(weight), ptr(lse), ptr(acc), c_i32(B), c_i32(h), c_i32(w),
         c_i32(H), c_i32(W), c_i32(nc), c_i32(ignore_index), stream_of(logits))
    return acc, lse


def ce_upsample_bwd(logits, label, weight, lse, gscale, gscale_mul, H, W, ignore_index=255, out=None):
    check_dev(logits, label, weight, lse, gscale, out)
    B, h, w, nc = logits.shape
    dl = torch.empty_like(logits) if out is None else out
    call('cmda_ce_upsample_bwd', ptr(logits), ptr(label), ptr(weight), ptr(lse), ptr(gscale), c_f32(gscale_mul), ptr(dl),
         c_i32(B), c_i32(h), c_i32(w), c_i32(H), c_i32(W), c_i32(nc), c_i32(ignore_index), stream_of(logits))
    return dl


def pseudo_label(logits, H, W, thr, want_prob=True):
    check_dev(logits)
    B, h, w, nc = logits.shape
    label = torch.empty(B, H, W, dtype=torch.int64, device=logits.device)
    prob = torch.empty(B, H, W, dtype=torch.float32, device=logits.device) if want_prob else None
    count = torch.zeros(1, dtype=torch.int32, device=logits.device)
    call('cmda_pseudo_label', ptr(logits), ptr(label), ptr(prob), ptr(count), c_i32(B), c_i32(h), c_i32(w), c_i32(H),
         c_i32(W), c_i32(nc), c_f32(thr), stream_of(logits))
    return label, prob, count


def pseudo_weight(count, B, H, W, top=0, bottom=0):
    check_dev(count)
    wgt = torch.empty(B, H, W, dtype=torch.float32, device=count.device)
    call('cmda_pseudo_weight', ptr(count), ptr(wgt), c_i32(B), c_i32(H), c_i32(W), c_i32(top), c_i32(bottom),
         stream_of(count))
    return wgt


def sample_scale(x, scale, B, C, per_channel=False, out=None):
    """x viewed as [B, HW*C]; scale fp32 [B] or [B,C]."""
    check_dev(x, scale, out)
    out = torch.empty_like(x) if out is None else out
    call('cmda_sample_scale', ptr(x), ptr(scale), ptr(out), c_i32(B), c_i64(x.numel() // B), c_i32(C),
         c_i32(int(per_channel)), dtype_tag(x), stream_of(x))
    return out


def upsample_logits_nchw(logits, H, W):
    """fp32 NHWC [B,h,w,nc] -> fp32 NCHW [B,nc,H,W] (bilinear, align_corners=False)."""
    check_dev(logits)
    B, h, w, nc = logits.shape
    out = torch.empty(B, nc, H, W, dtype=torch.float32, device=logits.device)
    call('cmda_upsample_logits_nchw', ptr(logits), ptr(out), c_i32(B), c_i32(h), c_i32(w), c_i32(H), c_i32(W), c_i32(nc),
         stream_of(logits))
    return out


def copy2d(src, dst, rows, cols, src_ld, dst_ld, src_off=0, dst_off=0):
    check_dev(src, dst)
    es = _ESIZE[src.dtype]
    call('cmda_copy2d', L.c_vp(src.data_ptr() + src_off * es), L.c_vp(dst.data_ptr() + dst_off * es), c_i64(rows),
         c_i32(cols), c_i64(src_ld), c_i64(dst_ld), dtype_tag(src), stream_of(src))
    return dst


_IMG_MEAN = (ctypes.c_float * 3)(123.675, 116.28, 103.53)
_IMG_STD = (ctypes.c_float * 3)(58.395, 57.12, 57.375)
_ISR_DIRS = {'rightdown': ((0, -1), (-1, 0)), 'rightup': ((0, -1), (1, 0)), 'leftdown': ((0, 1), (-1, 0)),
             'leftup': ((0, 1), (1, 0)), 'all': ((1, 0), (0, 1), (-1, 0), (0, -1))}


_ISR_LUT = {}


def isr_lut(val_range, device):
    """float32 log-intensity of the 256 gray levels, computed exactly as datasets/utils.py:get_ic does (numpy fp32); cached per
    (value range, device) -- the table is a constant of the configuration."""
    import numpy as np
    key = (float(val_range[0]), float(val_range[1]), str(device))
    lut = _ISR_LUT.get(key)
    if lut is None:
        g = np.arange(256, dtype=np.float32)
        lut = torch.from_numpy(np.log(g / 255 * (val_range[1] - val_range[0]) + val_range[0]).astype(np.float32)).to(device)
        _ISR_LUT[key] = lut
    return lut


def isr_dirs(shift_direction, shift_pixel):
    """host list of (dy, dx) shifts of get_image_change_from_pil (datasets/utils.py:108-152) for a direction name"""
    return [(dy * shift_pixel, dx * shift_pixel) for dy, dx in _ISR_DIRS[shift_direction]]


def isr_gray(img):
    """normalised NCHW fp32 image [B,3,H,W] -> PIL-exact 'L' uint8 [B,H,W]"""
    check_dev(img)
    B, _, H, W = img.shape
    gray = torch.empty(B, H, W, dtype=torch.uint8, device=img.device)
    call('cmda_isr_gray', ptr(img), ptr(gray), c_i32(B), c_i32(H), c_i32(W), _IMG_MEAN, _IMG_STD, stream_of(img))
    return gray


def isr_from_gray(gray, val_range, threshold, clip_range, shift_pixel, shift_direction, dirs_dev=None):
    """get_image_change_from_pil after the gray conversion -> fp32 NCHW [B,3,H,W] in [-1,1].  `dirs_dev`: optional DEVICE
    int32 [ndir,2] holding the (dy,dx) shifts (then `shift_direction` only gives ndir); lets a captured launch sequence pick
    the direction per replay."""
    import numpy as np
    check_dev(gray, dirs_dev)
    B, H, W = gray.shape
    span = np.log(val_range[1]) - np.log(val_range[0])
    dirs = isr_dirs(shift_direction, shift_pixel)
    dirs_t = dirs_dev if dirs_dev is not None else torch.tensor(dirs, dtype=torch.int32).to(gray.device)
    mm = torch.empty(B * len(dirs) * 4, dtype=torch.int32, device=gray.device)
    out = torch.empty(B, 3, H, W, dtype=torch.float32, device=gray.device)
    lut = isr_lut(val_range, gray.device)
    call('cmda_isr_from_gray', ptr(gray), ptr(lut), ptr(dirs_t), c_i32(len(dirs)), ptr(mm),
         ptr(out), c_i32(B), c_i32(H), c_i32(W), c_f32(float(np.float32(span * threshold))),
         c_f32(float(np.float32(span * clip_range))), stream_of(gray))
    return out


def events_to_voxel_grid(t, x, y, pol, bins, H, W):
    check_dev(t, x, y, pol)
    grid = torch.empty(bins, H, W, dtype=torch.float32, device=t.device)
    call('cmda_events_to_voxel_grid', ptr(t), ptr(x), ptr(y), ptr(pol), ptr(grid), c_i64(t.numel()), c_i32(bins), c_i32(H),
         c_i32(W), stream_of(t))
    return grid


def events_norm(events, clip_range, final_range=1.0):
    check_dev(events)
    out = torch.empty_like(events)
    ws = torch.empty(5, dtype=torch.float64, device=events.device)
    call('cmda_events_norm', ptr(events), ptr(out), ptr(ws), c_i64(events.numel()), c_f32(clip_range), c_f32(final_range),
         stream_of(events))
    return out


def jitter_params(per_sample):
    """host list of B tuples (order[4], f_b, f_c, f_s, f_h) -> CPU fp32 [B,8] (the layout cmda_color_jitter reads)"""
    return torch.tensor([[float(v) for v in order] + [fb, fc, fs, fh] for order, fb, fc, fs, fh in per_sample],
                        dtype=torch.float32)


def color_jitter_(img, prm, enable=None):
    """In place on the normalised NCHW fp32 image: kornia-0.5 ColorJitter, one (op order, factors) row of the DEVICE fp32
    tensor `prm` [B,8] per sample; `enable`: optional DEVICE int32 gate (0 = leave the image untouched)."""
    check_dev(img, prm, enable)
    B, _, H, W = img.shape
    assert prm.shape == (B, 8) and prm.dtype == torch.float32
    call('cmda_color_jitter', ptr(img), c_i32(B), c_i32(H), c_i32(W), _IMG_MEAN, _IMG_STD, ptr(prm), ptr(enable), stream_of(img))
    return img


def gaussian_taps(k, sigma, device=None):
    x = torch.arange(k, dtype=torch.float32) - k // 2
    g = torch.exp(-x * x / (2.0 * sigma * sigma))
    g = g / g.sum()
    return g if device is None else g.to(device)


def blur_kernel_size(n):
    """dacs_transforms.py:86-93: kernel size from the image extent n"""
    import numpy as np
    return int(np.floor(np.ceil(0.1 * n) - 0.5 + np.ceil(0.1 * n) % 2))


def gaussian_blur_(img, taps_x, taps_y, enable=None):
    """In place separable Gaussian blur (reflect border) of an NCHW fp32 image; taps_x / taps_y: DEVICE fp32 normalised
    Gaussians (lengths kx from W, ky from H); `enable`: optional DEVICE int32 gate."""
    check_dev(img, taps_x, taps_y, enable)
    B, C, H, W = img.shape
    tmp = torch.empty_like(img)
    call('cmda_gaussian_blur', ptr(img), ptr(tmp), ptr(taps_x), ptr(taps_y), c_i32(B * C), c_i32(H), c_i32(W),
         c_i32(taps_x.numel()), c_i32(taps_y.numel()), ptr(enable), stream_of(img))
    return img
