#!/usr/bin/env python3
"""Phase stamps of the bf16 lean GEMM (build with `make leantiming`, run with CMDA_HIP_LIB=build/libcmda_hip_leantiming.so): where the time of
workgroup 0 / wave 0 goes -- prologue, k-tiles, accumulators through LDS, rows stored.  python lean_phase.py M N K [nn] [res]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmda_amd import ops, _lib as L

M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 320, 320)
nn = 'nn' in sys.argv[4:]
bf = torch.bfloat16
a = torch.randn(M, K, device='cuda').to(bf)
b = (torch.randn(K, N, device='cuda') if nn else torch.randn(N, K, device='cuda')).to(bf)
bias = torch.randn(N, device='cuda')
res = torch.randn(M, N, device='cuda') if 'res' in sys.argv[4:] else None        # fp32 residual stream -> fp32 output
o = torch.empty(M, N, dtype=torch.float32 if res is not None else bf, device='cuda')
for _ in range(5):
    ops.gemm(ops.plain_view(a, M, K), ops.plain_view(b, K, N) if nn else ops.plain_view(b, N, K), o, M, N, K, dtype=1, b_kstrided=nn,
             bias=None if nn else bias, res=res)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
assert L.lib().cmda_debug_lean_stamps(buf) == 0
t = list(buf)
nkt = min(K // 64, 40)
cyc = lambda i: t[i] - t[0]
tops = [cyc(4 + i) for i in range(nkt)]
print(f'{M} x {N} x {K} {"NN" if nn else "NT"}{" + fp32 residual" if res is not None else ""}: workgroup 0 / wave 0, shader cycles from kernel entry')
print(f'  prologue issue done {cyc(1)}; k-tile tops {tops}; durations {[tops[i + 1] - tops[i] for i in range(nkt - 1)]} + last {cyc(50) - tops[-1]}')
print(f'  loop end {cyc(50)}, accumulators staged {cyc(51)}, row-loop iteration tops {cyc(53)} {cyc(55)}, rows stored {cyc(52)}')
