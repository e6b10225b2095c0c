#!/usr/bin/env python3
"""Where does a grouped weight-gradient launch spend its time?  Stage-3-like problem sets (MiT-B5: C = 320, hidden 1280) through
cmda_gemm_grouped with (a) the contraction length varied (slope = k-loop cost per k-tile, intercept = prologue + atomic epilogue),
(b) every block sharing ONE operand set (fits the Infinity Cache: tells HBM-bound from L2->LDS-bound), (c) the same list as
single cmda_gemm launches.  usage: grouped_dbg.py [nblocks]"""
import ctypes
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmda_amd import _lib as L, ops  # noqa: E402

dev = torch.device('cuda:0')
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 40


def make(tokens, nblocks, shared):
    """list of (GemmParams, flops) for nblocks transformer blocks: q, proj, fc1, fc2 (rows = tokens), kv (rows = tokens / 4)"""
    probs, keep = [], []
    C, Hd = 320, 1280
    pool = {}

    def t(rows, cols, tag, b):
        key = (rows, cols, tag) if shared else (rows, cols, tag, b)
        if key not in pool:
            pool[key] = torch.randn(rows, cols, device=dev).bfloat16()
        return pool[key]
    for b in range(nblocks):
        for (n_out, k_in, rows, tag) in ((C, C, tokens, 'q'), (C, C, tokens, 'proj'), (Hd, C, tokens, 'fc1'), (C, Hd, tokens, 'fc2'),
                                         (2 * C, C, tokens // 4, 'kv')):
            dy, x = t(rows, n_out, tag + 'dy', b), t(rows, k_in, tag + 'x', b)
            g = torch.zeros(n_out, k_in, dtype=torch.float32, device=dev)
            keep += [dy, x, g]
            p = L.GemmParams()
            p.A, p.B = ops.plain_view(dy, rows, n_out), ops.plain_view(x, rows, k_in)
            p.a_kstrided = p.b_kstrided = 1
            p.C, p.ldc = g.data_ptr(), k_in
            p.M, p.N, p.K, p.batch, p.batch2, p.splits = n_out, k_in, rows, 1, 1, 0
            p.alpha, p.beta, p.dtype, p.out_f32, p.atomic, p.c_vec_ok = 1.0, 0.0, 1, 1, 1, 1
            p.ldres = k_in
            p.rows_per_scale = 1
            probs.append((p, 2.0 * n_out * k_in * rows))
    return probs, keep


def time_it(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def run(tokens, nblocks, shared, label):
    probs, keep = make(tokens, nblocks, shared)
    n = len(probs)
    arr = (L.GemmParams * n)(*[p for p, _ in probs])
    flops = sum(f for _, f in probs)
    lib = L.lib()
    nbytes = int(lib.cmda_gemm_grouped_ws_bytes(arr, ctypes.c_int32(n)))
    host = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    dbuf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    st = L.stream_of(dbuf)
    L.call('cmda_gemm_grouped', arr, ctypes.c_int32(n), L.ptr(host), L.ptr(dbuf), ctypes.c_int64(nbytes), ctypes.c_int32(1), st)
    torch.cuda.synchronize()
    us_g = time_it(lambda: L.call('cmda_gemm_grouped', arr, ctypes.c_int32(n), L.ptr(host), L.ptr(dbuf), ctypes.c_int64(nbytes), ctypes.c_int32(0), st))

    def singles():
        for p, _ in probs:
            L.call('cmda_gemm', ctypes.byref(p), st)
    us_s = time_it(singles, reps=2)
    print(f'{label:34s} tokens {tokens:6d} blocks {nblocks:3d} problems {n:4d}: grouped {us_g:9.1f} us ({flops / us_g / 1e6:7.1f} TFLOP/s)   '
          f'single launches {us_s:9.1f} us ({flops / us_s / 1e6:7.1f} TFLOP/s)   map {nbytes // 1024} KB', flush=True)


for tokens in (512, 2048, 8192):
    run(tokens, NB, False, 'distinct operands')
run(32768, max(2, NB // 4), False, 'distinct operands')
run(8192, NB, True, 'ONE operand set (cache resident)')
run(2048, NB, True, 'ONE operand set (cache resident)')
