"""graph-timed cmda_ln_gemm against LayerNorm + Linear as two launches (us per dependent launch group)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmda_amd import ops


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


dev = torch.device('cuda:0')
for M, N, K in ((4096, 320, 320), (2048, 320, 320), (1024, 640, 320), (512, 640, 320), (65536, 64, 64), (16384, 128, 128)):
    x = torch.randn(M, K, device=dev)
    gamma, beta = torch.randn(K, device=dev), torch.randn(K, device=dev)
    w, b = (torch.randn(N, K, device=dev) * 0.1).bfloat16(), torch.randn(N, device=dev)
    xn = torch.empty(M, K, dtype=torch.bfloat16, device=dev)
    y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    h = ops.gemm(ops.plain_view(xn, M, K), ops.plain_view(w, N, K), y, M, N, K, dtype=1, bias=b, hold=True, keep=(xn,))
    res = {}
    for mode in (True, False):
        ops.LN_GEMM = mode
        res[mode] = timeit(lambda: ops.ln_gemm(x, gamma, beta, 1e-6, h))
    ops.LN_GEMM = True
    t_ln = timeit(lambda: ops.layernorm_fwd(x, gamma, beta, 1e-6, out=xn))
    t_g = timeit(lambda: ops.gemm(ops.plain_view(xn, M, K), ops.plain_view(w, N, K), y, M, N, K, dtype=1, bias=b))
    print(f'{M:6d} x {N:4d} x {K:4d}: fused {res[True]:6.2f} us   two launches {res[False]:6.2f} us   (LayerNorm {t_ln:5.2f}, Linear {t_g:5.2f})', flush=True)
