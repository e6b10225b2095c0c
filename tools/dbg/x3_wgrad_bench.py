#!/usr/bin/env python3
"""Graph-timed weight-gradient form of the lean split-bf16 GEMM (dW[N, K] += dy^T x over `tokens` rows) at the encoders' shapes;
isolated launches, one after the other."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmda_amd import ops  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from x3_bench import timeit  # noqa: E402

dev = torch.device('cuda:0')
tot = 0.0
for N, K, T in ((320, 320, 4096), (1280, 320, 4096), (320, 1280, 4096), (640, 320, 4096), (320, 320, 8192), (1280, 320, 8192), (64, 64, 65536), (256, 64, 65536),
                (128, 128, 16384), (512, 128, 16384), (512, 512, 1024), (2048, 512, 1024), (512, 512, 2048)):
    dy, x = torch.randn(T, N, device=dev), torch.randn(T, K, device=dev)
    dW, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    t = timeit(lambda: ops.gemm(ops.plain_view(dy, T, N), ops.plain_view(x, T, K), dW, N, K, T, a_kstrided=True, b_kstrided=True, dtype=2, atomic=True,
                                splits=0, colsum=db), iters=20)
    tot += t
    print(f'  dW[{N:4d} x {K:4d}] over {T:6d} tokens: {t:7.1f} us  {2.0 * N * K * T / t / 1e6:6.1f} TFLOP/s', flush=True)
print(f'  sum {tot:.1f} us')
