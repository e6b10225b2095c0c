#!/usr/bin/env python3
"""Graph-timed sweep of the second lean GEMM's tile / wave configurations (csrc/gemm_lean2.hip, tile_hint bit 15 + bits 16-21) against the
library's current choice, on the encoders' Linear shapes of the UDA step (stage 3: C = 320, hidden 1280; 2048 / 4096 / 8192 rows; the kv
Linear on a quarter of the rows; a few shapes of the other stages).  GPU box: python tools/dbg/lean2_bench.py [short]
(the experiment of round 6: output in profiles/r06_lean2_sweep.txt; needs the dispatch hook quoted in gemm_lean2.hip)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmda_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
CFG = {0: '64x64 8w s2', 1: '64x64 8w s3', 2: '64x64 4c+4p', 3: '64x80 4c+4p', 4: '64x80 10w', 5: '128x80 8w', 6: '128x80 4c+4p',
       7: '128x160 8w s2', 8: '128x160 8w s3', 9: '128x160 8c+4p', 10: '256x160 8w s2', 11: '256x160 8w s3', 12: '128x64 8w', 13: '128x128 8w',
       14: '128x128 8c+4p', 15: '64x80 4w', 16: '32x64 4c+4p', 17: '32x80 2c+2p'}
TILE = {0: (64, 64), 1: (64, 64), 2: (64, 64), 3: (64, 80), 4: (64, 80), 5: (128, 80), 6: (128, 80), 7: (128, 160), 8: (128, 160), 9: (128, 160),
        10: (256, 160), 11: (256, 160), 12: (128, 64), 13: (128, 128), 14: (128, 128), 15: (64, 80), 16: (32, 64), 17: (32, 80)}


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


def problem(M, N, K, res):
    a, b = torch.randn(M, K, device=dev).to(torch.bfloat16), (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    o = torch.empty(M, N, dtype=torch.float32 if res else torch.bfloat16, device=dev)
    r = torch.randn(M, N, device=dev) if res else None
    return a, b, bias, o, r


def run(pr, M, N, K, hint):
    a, b, bias, o, r = pr
    ops.GEMM_TILE_HINT = hint
    try:
        return timeit(lambda: ops.gemm(ops.plain_view(a, M, K), ops.plain_view(b, N, K), o, M, N, K, dtype=1, bias=bias, res=r))
    finally:
        ops.GEMM_TILE_HINT = 0


def check(pr, M, N, K, hint):
    a, b, bias, o, r = pr
    ops.GEMM_TILE_HINT = hint
    try:
        o.zero_()
        ops.gemm(ops.plain_view(a, M, K), ops.plain_view(b, N, K), o, M, N, K, dtype=1, bias=bias, res=r)
    finally:
        ops.GEMM_TILE_HINT = 0
    ref = a.float() @ b.float().t() + bias + (r if r is not None else 0)
    return ((o.float() - ref).abs().max() / ref.abs().max()).item()


_t = torch.zeros(256, device=dev)
print(f'dependent-launch floor (graph replay, trivial kernel): {timeit(lambda: ops.axpby(_t, _t, 1.0, 0.0, out=_t)):.2f} us per launch')
short = 'short' in sys.argv[1:]
shapes = []
for M in (2048, 4096, 8192):
    shapes += [(M, 320, 320, False), (M, 320, 320, True), (M, 1280, 320, False), (M, 320, 1280, True), (M // 4, 640, 320, False)]
if not short:
    shapes += [(32768, 64, 64, False), (65536, 64, 64, True), (65536, 256, 64, False), (65536, 64, 256, True),
               (16384, 128, 128, False), (16384, 512, 128, False), (16384, 128, 512, True), (32768, 512, 128, False),
               (1024, 512, 512, False), (1024, 2048, 512, False), (1024, 512, 2048, True), (2048, 2048, 512, False)]
cfgs = sorted(CFG)
print('us per launch (graph-timed, incl. the dependent-launch floor); * = best; (n workgroups)')
for M, N, K, res in shapes:
    pr = problem(M, N, K, res)
    t0 = run(pr, M, N, K, 0)
    row = []
    for c in cfgs:
        bm, bn = TILE[c]
        if bn > 2 * N:
            continue
        err = check(pr, M, N, K, 32768 | (c << 16))
        t = run(pr, M, N, K, 32768 | (c << 16))
        row.append((t, c, err, ((M + bm - 1) // bm) * ((N + bn - 1) // bn)))
    best = min(row)[0]
    fl = 2.0 * M * N * K
    print(f'{M:6d} x {N:5d} x {K:5d} {"+res32" if res else "      "}: library {t0:6.2f} us ({fl / t0 / 1e6:4.0f} TF) | best {best:6.2f} ({fl / best / 1e6:4.0f} TF, {t0 / best:4.2f}x)')
    print('      ' + '  '.join(f'{"*" if t == best else ""}[{c}] {CFG[c]} {t:.2f} ({nwg}){"" if err < 2e-2 else " ERR %.3g" % err}' for t, c, err, nwg in row))
