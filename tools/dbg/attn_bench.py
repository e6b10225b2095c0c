"""per-kernel times of the fused attention at the encoder's shapes (run under rocprofv3 --kernel-trace and read with
tools/trace_stats.py, or plain for event timings of forward / backward)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmda_amd import ops

def timeit(f, iters=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

bf = torch.bfloat16
for (B, N, heads) in ((4, 16384, 1), (2, 16384, 1), (4, 4096, 2), (4, 1024, 5), (2, 1024, 5), (4, 256, 8)):
    C, Nk = heads * 64, 256
    q = torch.randn(B * N, C, device='cuda').to(bf); kv = torch.randn(B * Nk, 2 * C, device='cuda').to(bf); do = torch.randn_like(q)
    direct = ops.attention_bwd_direct(B, N, Nk, heads)
    dkv32 = None if direct else torch.zeros(B * Nk, 2 * C, device='cuda')
    dkv16 = torch.empty(B * Nk, 2 * C, device='cuda', dtype=bf) if direct else None
    tf = timeit(lambda: ops.attention_fused_fwd(q, kv, B, N, Nk, heads, C, 0.125))
    tb = timeit(lambda: ops.attention_fused_bwd(q, kv, do, dkv32, B, N, Nk, heads, C, 0.125, dkv16=dkv16))
    print(f'B{B} N{N} heads{heads}: fwd {tf:6.1f} us   bwd (dq + dkv) {tb:6.1f} us  direct={direct}')
