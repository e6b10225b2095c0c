#!/usr/bin/env python3
"""Graph-timed GEMM tiles against the library heuristics on the C = 320 stage's Linear shapes, and the split-bf16 (CMDA_F32X3) kernel
against the exact-fp32 one on the step's large shapes.  GPU box: python tools/dbg/rp_bench.py
(round 4 also timed a 64 x 320 row-panel tile here -- measured slower than tile + LayerNorm launch, removed in round 5)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmda_amd import ops  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, iters=40, reps=5):
    """us per launch of `iters` dependent launches replayed from a hipGraph (eager launches from Python are host-bound at ~8 us)"""
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


def nt(M, N, K, hint, dt=torch.bfloat16, tag=1, res=False):
    a, b = torch.randn(M, K, device=dev).to(dt), torch.randn(N, K, device=dev).to(dt)
    bias = torch.randn(N, device=dev)
    o = torch.empty(M, N, dtype=torch.float32 if res else dt, device=dev)
    r = torch.randn(M, N, device=dev) if res else None
    ops.GEMM_TILE_HINT = hint
    t = timeit(lambda: ops.gemm(ops.plain_view(a, M, K), ops.plain_view(b, N, K), o, M, N, K, dtype=tag, bias=bias, res=r))
    ops.GEMM_TILE_HINT = 0
    return t


_t = torch.zeros(256, device=dev)
print(f'dependent-launch floor (graph replay, trivial kernel): {timeit(lambda: ops.axpby(_t, _t, 1.0, 0.0, out=_t)):.2f} us per launch')
_x = torch.randn(8192, 320, device=dev)
_g, _b = torch.ones(320, device=dev), torch.zeros(320, device=dev)
print(f'LayerNorm fwd fp32 -> bf16, 8192 x 320: {timeit(lambda: ops.layernorm_fwd(_x, _g, _b, 1e-6, out_dtype=torch.bfloat16)):.2f} us per launch')
if os.environ.get('RP_SHORT'):
    for M, N, K in ((2048, 320, 320), (8192, 320, 320), (8192, 1280, 320), (8192, 320, 1280)):
        print(f'  {M} x {N} x {K}: lean {nt(M, N, K, 0):.2f} us   general kernel {nt(M, N, K, 8192):.2f} us   general + general address path {nt(M, N, K, 8192 | 2048):.2f} us')
    sys.exit(0)
print('tiles vs heuristics, bf16, us per launch (back-to-back launches)')
for M in (2048, 4096, 8192, 16384):
    for N, K, res in ((320, 320, False), (320, 320, True), (640, 320, False), (1280, 320, False), (320, 1280, True)):
        t0 = nt(M, N, K, 0, res=res)
        tt = [nt(M, N, K, h, res=res) for h in (3, 3 | (4 << 4), 2, 2 | (4 << 4), 1, 1 | (4 << 4))]
        tg = [nt(M, N, K, h | 2048, res=res) for h in (3, 2, 1)]   # general address path
        fl = 2.0 * M * N * K
        print(f'  {M:6d} x {N:5d} x {K:5d} {"+res32" if res else "      "}: heuristics {t0:6.1f} us ({fl / t0 / 1e6:5.0f} TF) | 64x64 {tt[0]:5.1f} 4st {tt[1]:5.1f} | 128x64 {tt[2]:5.1f} 4st {tt[3]:5.1f} | '
              f'128x128 {tt[4]:5.1f} 4st {tt[5]:5.1f} | general path 64x64 {tg[0]:5.1f} 128x64 {tg[1]:5.1f} 128x128 {tg[2]:5.1f}')
print('split-bf16 (dtype 2) vs exact fp32 (dtype 0), fp32 storage')
for M, N, K in ((8192, 1280, 320), (8192, 320, 1280), (65536, 256, 1024), (16384, 1024, 1024), (4096, 4096, 4096)):
    t0 = nt(M, N, K, 0, torch.float32, 0)
    t2 = nt(M, N, K, 0, torch.float32, 2)
    fl = 2.0 * M * N * K
    print(f'  {M:6d} x {N:5d} x {K:5d}: exact {t0:8.1f} us ({fl / t0 / 1e6:6.0f} TF)   x3 {t2:8.1f} us ({fl / t2 / 1e6:6.0f} TF)')
