"""stage-3 encoder GEMMs (M = B x 1024 tokens, 320 / 1280 channels) alone: library heuristics against forced tiles"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmda_amd import ops

def timeit(fn, iters=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

bf = torch.bfloat16
dev = 'cuda'
r = lambda *s: torch.randn(*s, device=dev).to(bf)
z = torch.zeros(64, 8, device=dev)
print('floor (rows_fill tiny):', round(timeit(lambda: z.zero_()), 1))
shapes = [('k64', 'nt', 2048, 320, 64), ('k128', 'nt', 2048, 320, 128), ('k256', 'nt', 2048, 320, 256), ('k640', 'nt', 2048, 320, 640), ('k64 small', 'nt', 256, 64, 64), ('fc1 fwd', 'nt', 2048, 1280, 320), ('fc2 fwd', 'nt', 2048, 320, 1280), ('q fwd', 'nt', 2048, 320, 320), ('q fwd B4', 'nt', 4096, 320, 320),
          ('fc1 fwd B4', 'nt', 4096, 1280, 320), ('fc2 fwd B4', 'nt', 4096, 320, 1280),
          ('fc2 dgrad', 'nn', 2048, 1280, 320), ('fc1 dgrad', 'nn', 2048, 320, 1280), ('kv fwd', 'nt', 512, 640, 320),
          ('s2 fc1', 'nt', 8192, 512, 128), ('s2 fc2', 'nt', 8192, 128, 512), ('s1 fc1', 'nt', 32768, 256, 64), ('s1 fc2', 'nt', 32768, 64, 256)]
for name, kind, M, N, K in shapes:
    a = r(M, K)
    b = r(N, K) if kind == 'nt' else r(K, N)
    o = torch.empty(M, N, dtype=bf, device=dev)
    bias = torch.randn(N, device=dev)
    out = []
    for hint in (0, 3, 2, 1, 3 | 256):
        ops.GEMM_TILE_HINT = hint
        if kind == 'nt':
            f = lambda: ops.gemm(ops.plain_view(a, M, K), ops.plain_view(b, N, K), o, M, N, K, dtype=1, bias=bias)
        else:
            f = lambda: ops.gemm(ops.plain_view(a, M, K), ops.plain_view(b, K, N), o, M, N, K, b_kstrided=True, dtype=1)
        try:
            out.append(f'h{hint}: {timeit(f):5.1f}')
        except Exception as e:
            out.append(f'h{hint}: err')
    ops.GEMM_TILE_HINT = 0
    print(f'{name:12s} {M}x{N}x{K}: ' + '  '.join(out) + f'   ({2.0 * M * N * K / 1e9:.2f} GF)')
