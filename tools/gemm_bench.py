#!/usr/bin/env python3
"""Micro-benchmark of the GEMM kernel on the shapes the MiT-B5 + DAFormer step actually launches (per-GPU batch --batch, default 16).
Usage (GPU box): python tools/gemm_bench.py [--tile N]   -- prints us / TFLOP/s per shape; used for kernel tuning."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmda_amd import ops  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--big', action='store_true', help='only the large square-ish GEMMs, with and without the grouped tile walk')
    ap.add_argument('--hint', type=int, default=0, help='cmda_gemm_params_t.tile_hint for every launch (e.g. 1028 = force the ping-pong 256x256 kernel)')
    ap.add_argument('--splits', type=int, default=0, help='force the split-K count of the weight-gradient GEMMs')
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    tag = 1 if dt == torch.bfloat16 else 0
    dev = torch.device('cuda:0')
    ops.GEMM_TILE_HINT = args.hint
    r = lambda *s: torch.randn(*s, device=dev).to(dt)
    rows = []

    def nt(name, M, N, K):
        a, b, o = r(M, K), r(N, K), torch.empty(M, N, dtype=dt, device=dev)
        bias = torch.randn(N, device=dev)
        rows.append((name, 2.0 * M * N * K, timeit(lambda: ops.gemm(ops.plain_view(a, M, K), ops.plain_view(b, N, K), o, M, N, K, dtype=tag, bias=bias))))

    def nn(name, M, N, K):  # dx = dy[M,K] @ W[K,N]
        a, b, o = r(M, K), r(K, N), torch.empty(M, N, dtype=dt, device=dev)
        rows.append((name, 2.0 * M * N * K, timeit(lambda: ops.gemm(ops.plain_view(a, M, K), ops.plain_view(b, K, N), o, M, N, K, b_kstrided=True, dtype=tag))))

    def tn(name, M, N, K):  # dW[M,N] = dy[K,M]^T x[K,N]
        a, b, o = r(K, M), r(K, N), torch.zeros(M, N, device=dev)
        rows.append((name, 2.0 * M * N * K, timeit(lambda: ops.gemm(ops.plain_view(a, K, M), ops.plain_view(b, K, N), o, M, N, K, a_kstrided=True, b_kstrided=True, dtype=tag, atomic=True, splits=0))))

    def conv(name, B, H, W, Ci, Co, k):
        x, w = r(B * H * W, Ci), r(Co, k * k * Ci)
        M, K = B * H * W, k * k * Ci
        o = torch.empty(M, Co, dtype=dt, device=dev)
        rows.append((name + ' fwd', 2.0 * M * Co * K, timeit(lambda: ops.gemm(ops.conv_view(x, B, H, W, Ci, k, k, 1, k // 2), ops.plain_view(w, Co, K), o, M, Co, K, dtype=tag), 10)))
        dy, dw = r(M, Co), torch.zeros(Co, K, device=dev)
        rows.append((name + ' wgrad', 2.0 * M * Co * K, timeit(lambda: ops.gemm(ops.plain_view(dy, M, Co), ops.conv_view(x, B, H, W, Ci, k, k, 1, k // 2), dw, Co, K, M, a_kstrided=True, b_kstrided=True, dtype=tag, atomic=True, splits=0), 10)))

    if args.big:
        for hint, tag_ in ((0, 'library heuristics'), (1024 | 4, 'ping-pong 256^2 (gemm_pp.hip) forced'), (512 | 4, 'gemm_glds_kernel 256^2 forced'), (1, '128^2'), (3, '64^2')):
            ops.GEMM_TILE_HINT = hint
            rows.append((f'--- {tag_} (tile_hint {hint})', 0.0, 1.0))
            nt('NT 8192x8192x8192', 8192, 8192, 8192)
            nt('NT 4096x4096x4096', 4096, 4096, 4096)
            nt('NT 16384x4096x4096', 16384, 4096, 4096)
            nn('NN 8192x8192x8192', 8192, 8192, 8192)
            nt('NT 65536x1280x320', 65536, 1280, 320)
            nt('NT 65536x2048x512', 65536, 2048, 512)
            nt('NT 262144x256x1024 (head pw)', 262144, 256, 1024)
            nn('NN 262144x1024x256 (head dpw)', 262144, 1024, 256)
            conv('head bottleneck 3x3 1024->256 B16', 16, 128, 128, 1024, 256, 3)
            tn('TN 256x1024x262144 (head wpw)', 256, 1024, 262144)
        ops.GEMM_TILE_HINT = 0
        for name, fl, us in rows:
            print(f'{name:38s} {us:10.1f} us  {fl / us / 1e6:8.1f} TFLOP/s')
        return
    Bt = args.batch
    T1, T2, T3, T4 = Bt * 16384, Bt * 4096, Bt * 1024, Bt * 256
    nt(f's3 fc1   NT {T3}x1280x320', T3, 1280, 320)
    nt(f's3 fc2   NT {T3}x320x1280', T3, 320, 1280)
    nt(f's3 q     NT {T3}x320x320', T3, 320, 320)
    nn(f's3 dfc2  NN {T3}x1280x320', T3, 1280, 320)
    nn(f's3 dfc1  NN {T3}x320x1280', T3, 320, 1280)
    nn(f's3 dq    NN {T3}x320x320', T3, 320, 320)
    tn(f's3 wfc1  TN 1280x320x{T3}', 1280, 320, T3)
    tn(f's3 wfc2  TN 320x1280x{T3}', 320, 1280, T3)
    tn(f's3 wq    TN 320x320x{T3}', 320, 320, T3)
    nt(f's1 fc1   NT {T1}x256x64', T1, 256, 64)
    nt(f's1 fc2   NT {T1}x64x256', T1, 64, 256)
    tn(f's1 wfc1  TN 256x64x{T1}', 256, 64, T1)
    nt(f's2 fc1   NT {T2}x512x128', T2, 512, 128)
    nt(f's2 fc2   NT {T2}x128x512', T2, 128, 512)
    nt(f's4 fc1   NT {T4}x2048x512', T4, 2048, 512)
    nt(f's4 fc2   NT {T4}x512x2048', T4, 512, 2048)
    tn(f's4 wfc1  TN 2048x512x{T4}', 2048, 512, T4)
    nt(f'hd pw    NT {T1}x256x1024', T1, 256, 1024)
    nn(f'hd dpw   NN {T1}x1024x256', T1, 1024, 256)
    tn(f'hd wpw   TN 256x1024x{T1}', 256, 1024, T1)
    conv('hd bottleneck 3x3 1024->256', Bt, 128, 128, 1024, 256, 3)
    nt('big      NT 8192x8192x8192', 8192, 8192, 8192)
    tot = 0
    for name, fl, us in rows:
        print(f'{name:38s} {us:10.1f} us  {fl / us / 1e6:8.1f} TFLOP/s')
    print('env', {k: v for k, v in os.environ.items() if k.startswith('CMDA_')})


if __name__ == '__main__':
    main()
