#!/usr/bin/env python3
"""Per-segment cycle counts of the ping-pong GEMM (build: `make pptiming`, run with CMDA_HIP_LIB=build/libcmda_hip_pptiming.so).
Prints, for wave 0 (G0) and wave 4 (G1) of one workgroup and k-tiles 8..11, the s_memtime deltas of every phase:
LOAD (segment start -> barrier passed + reads retired), MFMA issue (-> 8 MFMAs issued), tail (-> closing barrier passed)."""
import ctypes
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmda_amd import _lib as L, ops  # noqa: E402

dev = torch.device('cuda:0')
M = N = K = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
mode = sys.argv[2] if len(sys.argv) > 2 else 'nt'
a = torch.randn(M, K, device=dev).bfloat16()
b = torch.randn(N, K, device=dev).bfloat16() if mode == 'nt' else torch.randn(K, N, device=dev).bfloat16()
o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
ops.GEMM_TILE_HINT = 4 | 1024
for _ in range(3):
    if mode == 'nt':
        ops.gemm(ops.plain_view(a, M, K), ops.plain_view(b, N, K), o, M, N, K, dtype=1)
    else:
        ops.gemm(ops.plain_view(a, M, K), ops.plain_view(b, K, N), o, M, N, K, dtype=1, b_kstrided=True)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
assert L.lib().cmda_debug_pp_stamps(buf) == 0
st = list(buf)
for g in range(2):
    print(f'group G{g} (wave {4 * g}):  per phase q: LOAD | MFMA issue | tail-to-barrier   [cycles]')
    for t in range(4):
        row = []
        for q in range(4):
            s = st[((g * 4 + t) * 4 + q) * 4:((g * 4 + t) * 4 + q) * 4 + 4]
            row.append(f'{s[1] - s[0]:5d} {s[2] - s[1]:5d} {s[3] - s[2]:5d}')
        tile = st[((g * 4 + t) * 4 + 3) * 4 + 3] - st[((g * 4 + t) * 4) * 4]
        print(f'  k-tile {8 + t}: ' + ' | '.join(row) + f'   = {tile} per k-tile')
