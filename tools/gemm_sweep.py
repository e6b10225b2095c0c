#!/usr/bin/env python3
"""Forced (tile, stages) sweep of the LDS-DMA GEMM over the DACS step's own plain-view shapes (bench.py CMDA_BENCH_GEMM_HIST):
per shape the time of every configuration, 40 back-to-back launches between two events (launch floor included: compare rows,
not absolute numbers).  usage: python tools/gemm_sweep.py [hist.txt]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmda_amd import ops  # noqa: E402

SHAPES = [  # M N K a_kstrided b_kstrided atomic   (stage-3 Linears of the three encoder batches, their dgrads and wgrads)
    (2048, 320, 320, 0, 0, 0), (4096, 320, 320, 0, 0, 0), (8192, 320, 320, 0, 0, 0),
    (2048, 1280, 320, 0, 0, 0), (4096, 1280, 320, 0, 0, 0), (8192, 1280, 320, 0, 0, 0),
    (2048, 320, 1280, 0, 0, 0), (4096, 320, 1280, 0, 0, 0), (8192, 320, 1280, 0, 0, 0),
    (4096, 320, 1280, 0, 1, 0), (8192, 320, 1280, 0, 1, 0), (4096, 1280, 320, 0, 1, 0), (8192, 1280, 320, 0, 1, 0),
    (4096, 320, 320, 0, 1, 0), (8192, 320, 320, 0, 1, 0),
    (512, 640, 320, 0, 0, 0), (1024, 640, 320, 0, 0, 0), (2048, 640, 320, 0, 0, 0),
    (320, 1280, 8192, 1, 1, 1), (320, 320, 8192, 1, 1, 1), (1280, 320, 8192, 1, 1, 1), (320, 320, 4096, 1, 1, 1),
    (1280, 320, 4096, 1, 1, 1), (640, 320, 2048, 1, 1, 1),
    (16384, 128, 128, 0, 0, 0), (16384, 512, 128, 0, 0, 0), (16384, 128, 512, 0, 0, 0),
    (65536, 64, 64, 0, 0, 0), (65536, 256, 64, 0, 0, 0), (65536, 64, 256, 0, 0, 0),
    (1024, 512, 512, 0, 0, 0), (1024, 2048, 512, 0, 0, 0), (1024, 512, 2048, 0, 0, 0),
]
CONFIGS = [('auto', 0)] + [(f't{t}/ns{ns}', (t + 1) | (ns << 4)) for t, nss in ((2, (2, 4)), (1, (2, 4)), (0, (2, 4)))
                           for ns in nss]


def main():
    dev = torch.device('cuda:0')
    r = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
    print('%-34s' % 'M N K aks bks atomic' + ''.join('%10s' % n for n, _ in CONFIGS))
    for M, N, K, aks, bks, atomic in SHAPES:
        a = r(K, M) if aks else r(M, K)
        b = r(K, N) if bks else r(N, K)
        o = torch.zeros(M, N, dtype=torch.float32 if atomic else torch.bfloat16, device=dev)
        av = ops.plain_view(a, *a.shape)
        bv = ops.plain_view(b, *b.shape)
        row = []
        for name, hint in CONFIGS:
            ops.GEMM_TILE_HINT = hint

            def run():
                ops.gemm(av, bv, o, M, N, K, a_kstrided=bool(aks), b_kstrided=bool(bks), dtype=1, atomic=bool(atomic),
                         splits=0 if atomic else 1)
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                run()
            e1.record()
            torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / 40 * 1e3)
        best = min(row)
        print('%-34s' % f'{M} {N} {K} {aks} {bks} {atomic}' + ''.join(('%9.1f%s' % (v, '*' if v == best else ' ')) for v in row), flush=True)
    ops.GEMM_TILE_HINT = 0


if __name__ == '__main__':
    main()
