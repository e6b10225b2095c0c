"""GPU-side durations of small GEMMs: run under rocprofv3 --kernel-trace and read with tools/trace_stats.py --runs"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmda_amd import ops
bf = torch.bfloat16
r = lambda *s: torch.randn(*s, device='cuda').to(bf)
shapes = [(256, 64, 64), (2048, 320, 64), (2048, 320, 320), (2048, 320, 640), (2048, 320, 1280), (2048, 1280, 320), (4096, 320, 320), (512, 640, 320), (8192, 512, 128), (32768, 256, 64)]
hints = [int(h) for h in sys.argv[1:]] or [0]
for M, N, K in [(m, n, k) for (m, n, k) in shapes for _ in hints]:
    pass
for (M, N, K), hint in [((m, n, k), h) for (m, n, k) in shapes for h in hints]:
    ops.GEMM_TILE_HINT = hint
    a, b, o = r(M, K), r(N, K), torch.empty(M, N, dtype=bf, device='cuda')
    bias = torch.randn(N, device='cuda')
    for _ in range(30):
        ops.gemm(ops.plain_view(a, M, K), ops.plain_view(b, N, K), o, M, N, K, dtype=1, bias=bias)
    torch.cuda.synchronize()
    z = torch.zeros(8, device='cuda')   # separator kernel
    torch.cuda.synchronize()
