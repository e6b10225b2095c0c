#!/usr/bin/env python3
"""Phase timing of the LDS-DMA GEMM kernel (needs `make timing`; run with CMDA_HIP_LIB=build/libcmda_hip_timing.so).
Per shape: median over the first 256 blocks of the wall-clock (100 MHz) deltas between
  0 start | 1 address setup + prologue issue | 2 first stage landed | 3 k-loop done | 4 accumulators staged in LDS | 5 stored
and the spread of block start times (how the grid is scheduled)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmda_amd import _lib, ops  # noqa: E402


def stamps():
    buf = (ctypes.c_ulonglong * (256 * 8))()
    assert _lib.lib().cmda_debug_gemm_stamps(buf) == 0
    return np.frombuffer(buf, dtype=np.uint64).reshape(256, 8).astype(np.int64)


def run(name, fn, nblocks):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = stamps()[:min(256, nblocks), :6]
    d = np.diff(s, axis=1) * 10.0 / 1e3  # us
    med = np.median(d, axis=0)
    start = (s[:, 0] - s[:, 0].min()) * 10.0 / 1e3
    end = (s[:, 5] - s[:, 0].min()) * 10.0 / 1e3
    print(f'{name:34s} setup {med[0]:5.2f}  first-stage {med[1]:5.2f}  k-loop {med[2]:6.2f}  stage-C {med[3]:5.2f}  store {med[4]:5.2f} '
          f'| block total {np.median(end - start):6.2f}  start spread {start.max():6.2f}  last end {end.max():6.2f} us')


def main():
    dev = torch.device('cuda:0')
    dt, tag = torch.bfloat16, 1
    r = lambda *s: torch.randn(*s, device=dev).to(dt)

    def nt(name, M, N, K):
        a, b, o = r(M, K), r(N, K), torch.empty(M, N, dtype=dt, device=dev)
        bias = torch.randn(N, device=dev)
        run(name, lambda: ops.gemm(ops.plain_view(a, M, K), ops.plain_view(b, N, K), o, M, N, K, dtype=tag, bias=bias), 1 << 30)

    def nn(name, M, N, K):
        a, b, o = r(M, K), r(K, N), torch.empty(M, N, dtype=dt, device=dev)
        run(name, lambda: ops.gemm(ops.plain_view(a, M, K), ops.plain_view(b, K, N), o, M, N, K, b_kstrided=True, dtype=tag), 1 << 30)

    # the DACS step's own shapes (2 + 2 samples per GPU: the event encoder sees 4 x 1024 stage-3 tokens, the image encoder 2 x 1024)
    nt('s3 q    NT 4096x320x320 (B=4)', 4096, 320, 320)
    nt('s3 q    NT 2048x320x320 (B=2)', 2048, 320, 320)
    nt('s3 fc1  NT 4096x1280x320', 4096, 1280, 320)
    nt('s3 fc2  NT 4096x320x1280', 4096, 320, 1280)
    nn('s3 dfc1 NN 4096x320x1280', 4096, 320, 1280)
    nn('s3 dq   NN 4096x320x320', 4096, 320, 320)
    nt('s1 fc1  NT 65536x256x64', 65536, 256, 64)
    nt('s2 fc1  NT 16384x512x128', 16384, 512, 128)
    nt('s4 fc2  NT 1024x512x2048', 1024, 512, 2048)
    nt('s3 q    NT 16384x320x320', 16384, 320, 320)
    nt('s3 fc1  NT 16384x1280x320', 16384, 1280, 320)
    nt('s3 fc2  NT 16384x320x1280', 16384, 320, 1280)
    nn('s3 dfc1 NN 16384x320x1280', 16384, 320, 1280)
    nt('s1 fc1  NT 262144x256x64', 262144, 256, 64)
    nt('s2 fc1  NT 65536x512x128', 65536, 512, 128)
    nn('hd dpw  NN 262144x1024x256', 262144, 1024, 256)
    nt('hd pw   NT 262144x256x1024', 262144, 256, 1024)
    nt('big     NT 8192x8192x8192', 8192, 8192, 8192)
    print('env', {k: v for k, v in os.environ.items() if k.startswith('CMDA_')})


if __name__ == '__main__':
    main()
