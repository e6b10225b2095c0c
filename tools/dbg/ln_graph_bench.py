#!/usr/bin/env python3
"""Graph-timed LayerNorm forward / backward at the encoders' shapes (fp32 stream -> bf16, us per dependent launch); CMDA_LN_FWD_WIDE=1:
the 64-lane row groups of rounds 1-4 for C = 320 (A/B)."""
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


for rows, C in ((2048, 320), (4096, 320), (8192, 320), (65536, 64), (16384, 128), (1024, 512), (32768, 64), (8192, 128)):
    x = torch.randn(rows, C, device=dev)
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    y = torch.empty(rows, C, dtype=torch.bfloat16, device=dev)
    t_f = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-6, out=y))
    _, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6, out=y)
    dy = torch.randn(rows, C, device=dev).bfloat16()
    dres = torch.randn(rows, C, device=dev).bfloat16()
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    t_b = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dg, db, dres=dres))
    print(f'{rows:6d} x {C:4d}: fwd {t_f:6.2f} us   bwd {t_b:6.2f} us', flush=True)
