#!/usr/bin/env python3
"""Graph-timed split-bf16 (CMDA_F32X3) GEMMs at the encoders' Linear / data-gradient shapes: the LDS-DMA lean instance
(csrc/gemm_x3_lean.hip) against the register-staged general kernel (tile_hint bit 13) and the bf16 lean kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmda_amd import ops  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, iters=40, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * reps) * 1e3


def run(M, N, K, nn, hint, dt, tag):
    a = torch.randn(M, K, device=dev).to(dt)
    b = (torch.randn(K, N, device=dev) if nn else torch.randn(N, K, device=dev)).to(dt)
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    o = torch.empty(M, N, dtype=torch.float32, device=dev)
    ops.GEMM_TILE_HINT = hint

    def f():
        ops.gemm(ops.plain_view(a, M, K), ops.plain_view(b, K, N) if nn else ops.plain_view(b, N, K), o, M, N, K, dtype=tag, bias=None if nn else bias,
                 res=None if nn else res, b_kstrided=nn)
    t = timeit(f)
    ops.GEMM_TILE_HINT = 0
    return t, o.clone(), (a, b, bias, res)


def sr_forms():
    """the spatial-reduction convolution of stage 3 (2 x 2 / stride 2 on 32 x 32 tokens, C = 320) at B = 2 / 4 / 8 samples: forward with
    split-K atomics (patch view as A), weight gradient (patch view as K-strided B), data gradient with the patch-store epilogue"""
    print('spatial-reduction convolution forms, us per launch: library choice | register-staged general kernel (tile_hint bit 13)')
    C, k, HW = 320, 2, 32
    for B in (2, 4, 8):
        x = torch.randn(B, HW, HW, C, device=dev)
        w = torch.randn(C, k * k * C, device=dev)
        M, K = B * (HW // k) ** 2, k * k * C
        out = torch.zeros(M, C, device=dev)
        dy = torch.randn(M, C, device=dev)
        dW = torch.zeros(C, K, device=dev)
        dx = torch.empty(B * HW * HW, C, device=dev)
        forms = {
            'fwd split-K 5': lambda: ops.gemm(ops.conv_view(x, B, HW, HW, C, k, k, k, 0, 1), ops.plain_view(w, C, K), out, M, C, K, dtype=2, atomic=True, splits=5),
            'wgrad (patch B)': lambda: ops.gemm(ops.plain_view(dy, M, C), ops.conv_view(x, B, HW, HW, C, k, k, k, 0, 1), dW, C, K, M, a_kstrided=True, b_kstrided=True,
                                                dtype=2, atomic=True, splits=4),
            'dgrad patch-store': lambda: ops.gemm(ops.plain_view(dy, M, C), ops.plain_view(w, C, K), dx, M, K, C, b_kstrided=True, dtype=2, c_patch=(HW // k, k, k * C)),
        }
        for name, f in forms.items():
            ts = []
            for hint in (0, 8192):
                ops.GEMM_TILE_HINT = hint
                ts.append(timeit(f))
            ops.GEMM_TILE_HINT = 0
            print(f'  B = {B}  {name:18s} {ts[0]:7.1f} | {ts[1]:7.1f}', flush=True)


def short_k():
    """shapes where the tile heuristics choose 128 x 128 (register-staged kernel) although the lean kernel takes them: stage-1 / 2 Linears
    and the unfused attention's batched products; library choice | lean forced (bit 16) | general forced (bit 13)"""
    print('short-K shapes, us per launch: library choice | lean forced | general forced')
    for M, N, K, nb in ((32768, 256, 64, 1), (65536, 256, 64, 1), (131072, 256, 64, 1), (16384, 512, 128, 1), (32768, 512, 128, 1), (65536, 128, 256, 1),
                        (4096, 256, 64, 16), (16384, 256, 64, 4), (16384, 256, 64, 8)):
        a = torch.randn(nb, M, K, device=dev)
        b = torch.randn(nb, N, K, device=dev)
        o = torch.empty(nb, M, N, device=dev)
        ts = []
        for hint in (0, 65536, 8192):
            ops.GEMM_TILE_HINT = hint
            ts.append(timeit(lambda: ops.gemm(ops.plain_view(a, M, K, batch_stride=M * K), ops.plain_view(b, N, K, batch_stride=N * K), o, M, N, K, batch=nb,
                                              c_batch_stride=M * N, dtype=2), iters=20))
        ops.GEMM_TILE_HINT = 0
        print(f'  {M:6d} x {N:4d} x {K:4d} batch {nb:2d}: {ts[0]:7.1f} | {ts[1]:7.1f} | {ts[2]:7.1f}', flush=True)


if __name__ == "__main__":
    sr_forms()
    short_k()
    print('split-bf16 GEMMs, us per launch: lean (LDS-DMA) | general (register-staged) | bf16 lean kernel;  max |lean - general| / max|general|')
    for nn in (False, True):
        for M, N, K in ((2048, 320, 320), (4096, 320, 320), (8192, 320, 320), (8192, 1280, 320), (8192, 320, 1280), (4096, 640, 320),
                        (16384, 128, 128), (16384, 512, 128), (65536, 64, 64), (1024, 512, 512), (1024, 2048, 512),
                        (262144, 256, 1024), (262144, 1024, 256), (131072, 256, 256), (65536, 256, 512), (4096, 4096, 4096))[int(os.environ.get('X3_FROM', 0)):]:
            torch.manual_seed(M + N)
            t_lean, o_lean, _ = run(M, N, K, nn, int(os.environ.get('X3_LEAN_HINT', 0)), torch.float32, 2)
            torch.manual_seed(M + N)
            t_gen, o_gen, _ = run(M, N, K, nn, 8192, torch.float32, 2)
            t_bf, _, _ = run(M, N, K, nn, 0, torch.bfloat16, 1)
            err = (o_lean - o_gen).abs().max().item() / o_gen.abs().max().item()
            print(f'  {"NN" if nn else "NT"} {M:6d} x {N:5d} x {K:5d}: {t_lean:7.1f} | {t_gen:7.1f} | {t_bf:6.1f}   err {err:.1e}', flush=True)
