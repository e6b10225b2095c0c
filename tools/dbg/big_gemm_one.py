"""one large GEMM a few times (for rocprofv3 --pmc runs): python big_gemm_one.py M N K [hint]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmda_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4])
ops.GEMM_TILE_HINT = int(sys.argv[4]) if len(sys.argv) > 4 else 0
bf = torch.bfloat16
a, b = torch.randn(M, K, device='cuda').to(bf), torch.randn(N, K, device='cuda').to(bf)
o = torch.empty(M, N, dtype=bf, device='cuda')
for _ in range(4):
    ops.gemm(ops.plain_view(a, M, K), ops.plain_view(b, N, K), o, M, N, K, dtype=1)
torch.cuda.synchronize()
