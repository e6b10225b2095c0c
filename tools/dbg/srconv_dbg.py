"""SR-conv (kernel == stride) GEMMs of the step: conv view vs the equivalent plain GEMM, weight gradient with / without the
permuted atomic store (c_perm)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmda_amd import ops
dev = torch.device('cuda:0')
bf = torch.bfloat16


def timeit(fn, iters=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (B, H, C, s) in ((2, 32, 320, 2), (4, 32, 320, 2), (8, 32, 320, 2), (8, 64, 128, 4), (8, 128, 64, 8)):
    W = H
    OH = OW = H // s
    M, K, Co = B * OH * OW, s * s * C, C
    x = torch.randn(B * H * W, C, device=dev).to(bf)
    w = torch.randn(Co, K, device=dev).to(bf)
    bias = torch.randn(Co, device=dev)
    y = torch.empty(M, Co, dtype=bf, device=dev)
    xp = torch.randn(M, K, device=dev).to(bf)
    t_conv = timeit(lambda: ops.gemm(ops.conv_view(x, B, H, W, C, s, s, s, 0, 1, OH=OH, OW=OW), ops.plain_view(w, Co, K), y, M, Co, K, dtype=1, bias=bias))
    t_plain = timeit(lambda: ops.gemm(ops.plain_view(xp, M, K), ops.plain_view(w, Co, K), y, M, Co, K, dtype=1, bias=bias))
    dy = torch.randn(M, Co, device=dev).to(bf)
    dw = torch.zeros(Co, K, device=dev)
    t_wp = timeit(lambda: ops.gemm(ops.plain_view(dy, M, Co), ops.conv_view(x, B, H, W, C, s, s, s, 0, 1, OH=OH, OW=OW), dw, Co, K, M,
                                   a_kstrided=True, b_kstrided=True, dtype=1, atomic=True, splits=0, c_perm=(C, s * s)))
    t_wn = timeit(lambda: ops.gemm(ops.plain_view(dy, M, Co), ops.conv_view(x, B, H, W, C, s, s, s, 0, 1, OH=OH, OW=OW), dw, Co, K, M,
                                   a_kstrided=True, b_kstrided=True, dtype=1, atomic=True, splits=0))
    t_wplain = timeit(lambda: ops.gemm(ops.plain_view(dy, M, Co), ops.plain_view(xp, M, K), dw, Co, K, M,
                                       a_kstrided=True, b_kstrided=True, dtype=1, atomic=True, splits=0))
    print(f'B{B} H{H} C{C} s{s}: M {M} N {Co} K {K} | fwd conv-view {t_conv:6.1f} us, plain {t_plain:6.1f} | wgrad c_perm {t_wp:6.1f}, [Co,K] store {t_wn:6.1f}, plain operands {t_wplain:6.1f}')
