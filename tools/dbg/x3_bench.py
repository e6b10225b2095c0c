#!/usr/bin/env python3
"""Graph-timed split-bf16 (CMDA_F32X3) GEMMs at the encoders' Linear / data-gradient shapes: the LDS-DMA lean instance
(csrc/gemm_x3_lean.hip) against the register-staged general kernel (tile_hint bit 13) and the bf16 lean kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmda_amd import ops  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, iters=40, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * reps) * 1e3


def run(M, N, K, nn, hint, dt, tag):
    a = torch.randn(M, K, device=dev).to(dt)
    b = (torch.randn(K, N, device=dev) if nn else torch.randn(N, K, device=dev)).to(dt)
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    o = torch.empty(M, N, dtype=torch.float32, device=dev)
    ops.GEMM_TILE_HINT = hint

    def f():
        ops.gemm(ops.plain_view(a, M, K), ops.plain_view(b, K, N) if nn else ops.plain_view(b, N, K), o, M, N, K, dtype=tag, bias=None if nn else bias,
                 res=None if nn else res, b_kstrided=nn)
    t = timeit(f)
    ops.GEMM_TILE_HINT = 0
    return t, o.clone(), (a, b, bias, res)


print('split-bf16 GEMMs, us per launch: lean (LDS-DMA) | general (register-staged) | bf16 lean kernel;  max |lean - general| / max|general|')
for nn in (False, True):
    for M, N, K in ((2048, 320, 320), (4096, 320, 320), (8192, 320, 320), (8192, 1280, 320), (8192, 320, 1280), (4096, 640, 320),
                    (16384, 128, 128), (16384, 512, 128), (65536, 64, 64), (1024, 512, 512), (1024, 2048, 512),
                    (262144, 256, 1024), (262144, 1024, 256), (131072, 256, 256), (65536, 256, 512), (4096, 4096, 4096))[int(os.environ.get('X3_FROM', 0)):]:
        torch.manual_seed(M + N)
        t_lean, o_lean, _ = run(M, N, K, nn, int(os.environ.get('X3_LEAN_HINT', 0)), torch.float32, 2)
        torch.manual_seed(M + N)
        t_gen, o_gen, _ = run(M, N, K, nn, 8192, torch.float32, 2)
        t_bf, _, _ = run(M, N, K, nn, 0, torch.bfloat16, 1)
        err = (o_lean - o_gen).abs().max().item() / o_gen.abs().max().item()
        print(f'  {"NN" if nn else "NT"} {M:6d} x {N:5d} x {K:5d}: {t_lean:7.1f} | {t_gen:7.1f} | {t_bf:6.1f}   err {err:.1e}', flush=True)
