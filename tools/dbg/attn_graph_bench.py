"""graph-timed (dependent launches replayed from a hipGraph) fused attention kernels at the encoder's stage shapes"""
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


bf = torch.bfloat16
for (B, N, heads) in ((4, 16384, 1), (2, 16384, 1), (4, 4096, 2), (2, 4096, 2), (4, 1024, 5), (2, 1024, 5), (4, 256, 8), (2, 256, 8)):
    C, Nk = heads * 64, 256
    q = torch.randn(B * N, C, device='cuda').to(bf); kv = torch.randn(B * Nk, 2 * C, device='cuda').to(bf); do = torch.randn_like(q)
    direct = ops.attention_bwd_direct(B, N, Nk, heads)
    dkv32 = None if direct else torch.zeros(B * Nk, 2 * C, device='cuda')
    dkv16 = torch.empty(B * Nk, 2 * C, device='cuda', dtype=bf) if direct else None
    o = torch.empty_like(q)
    tf = timeit(lambda: ops.attention_fused_fwd(q, kv, B, N, Nk, heads, C, 0.125))
    tb = timeit(lambda: ops.attention_fused_bwd(q, kv, do, dkv32, B, N, Nk, heads, C, 0.125, dkv16=dkv16))
    print(f'B{B} N{N} heads{heads}: fwd {tf:6.1f} us   bwd (dq + dkv) {tb:6.1f} us  direct={direct}', flush=True)
