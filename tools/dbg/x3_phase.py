#!/usr/bin/env python3
"""Phase stamps of the lean split-bf16 GEMM (build with `make x3timing`, run with CMDA_HIP_LIB=build/libcmda_hip_x3timing.so): where the time
of workgroup 0 / wave 0 goes -- prologue, first tile's latency, k-tiles, epilogue.  python x3_phase.py M N K [nn]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmda_amd import ops, _lib as L

M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 320, 320)
nn = len(sys.argv) > 4 and sys.argv[4] == 'nn'
a = torch.randn(M, K, device='cuda')
b = torch.randn(K, N, device='cuda') if nn else torch.randn(N, K, device='cuda')
bias, res = torch.randn(N, device='cuda'), torch.randn(M, N, device='cuda')
o = torch.empty(M, N, device='cuda')
for _ in range(5):
    ops.gemm(ops.plain_view(a, M, K), ops.plain_view(b, K, N) if nn else ops.plain_view(b, N, K), o, M, N, K, dtype=2, b_kstrided=nn,
             bias=None if nn else bias, res=None if nn else res)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
assert L.lib().cmda_debug_x3_stamps(buf) == 0
t = list(buf)
nkt = min(K // 32, 40)
t0 = t[0]
cyc = lambda i: t[i] - t0   # s_memtime ticks = shader cycles
print(f'{M} x {N} x {K} {"NN" if nn else "NT"}: workgroup 0 / wave 0, cycles from kernel entry')
print(f'  prologue issue done {cyc(1)}, tile 0 landed {cyc(2)}, first split done {cyc(3)}')
tops = [cyc(4 + i) for i in range(nkt)]
print('  k-tile tops:', tops)
print('  k-tile durations:', [tops[i + 1] - tops[i] for i in range(nkt - 1)], '-> loop end', cyc(50) - tops[-1])
print(f'  loop end {cyc(50)}, accumulators staged {cyc(51)}, epilogue done {cyc(52)}')
