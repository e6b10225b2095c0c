"""one split-bf16 GEMM a few times (for rocprofv3 --pmc runs): python x3_gemm_one.py M N K [hint] [nn]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmda_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4])
ops.GEMM_TILE_HINT = int(sys.argv[4]) if len(sys.argv) > 4 else 0
nn = len(sys.argv) > 5 and sys.argv[5] == 'nn'
a = torch.randn(M, K, device='cuda')
b = torch.randn(K, N, device='cuda') if nn else torch.randn(N, K, device='cuda')
o = torch.empty(M, N, device='cuda')
for _ in range(6):
    ops.gemm(ops.plain_view(a, M, K), ops.plain_view(b, K, N) if nn else ops.plain_view(b, N, K), o, M, N, K, dtype=2, b_kstrided=nn)
torch.cuda.synchronize()
