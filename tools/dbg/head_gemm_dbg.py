"""decode-head pointwise GEMMs (M = 262144 rows) under forced tiles: which tile should the heuristic pick for short K?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmda_amd import ops
dev = torch.device('cuda:0')
bf = torch.bfloat16


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


M = 262144
for (N, K, bks) in ((1024, 256, True), (1024, 256, False), (256, 1024, False), (256, 256, False), (256, 320, False), (256, 512, False), (19, 256, False)):
    a = torch.randn(M, K, device=dev).to(bf)
    b = (torch.randn(K, N, device=dev) if bks else torch.randn(N, K, device=dev)).to(bf)
    o = torch.empty(M, N, dtype=bf, device=dev)
    row = []
    for hint in (0, 4, 1, 2, 3):
        ops.GEMM_TILE_HINT = hint
        row.append(timeit(lambda: ops.gemm(ops.plain_view(a, M, K), ops.plain_view(b, *b.shape), o, M, N, K, b_kstrided=bks, dtype=1)))
    ops.GEMM_TILE_HINT = 0
    fl = 2.0 * M * N * K
    print(f'M {M} N {N} K {K} {"NN" if bks else "NT"}: ' + '  '.join(f'{n} {t:7.1f} us ({fl / t / 1e6:5.0f} TF)' for n, t in zip(('auto', '256^2', '128^2', '128x64', '64^2'), row)))
