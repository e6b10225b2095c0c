"""Micro-benchmark of the HBM-bound kernels at the MiT-B5 512x512 stage shapes for a given encoder batch (the DACS step at
2 + 2 samples: 8 = event encoder over events + ISR of the source and the mixed samples, 4 = image encoder, 2 = teacher).

    python tools/hbm_bench.py [--batch 8]
prints per kernel: avg microseconds, algorithmic GB/s (DESIGN.md section 5 byte counts) and the fraction of 8 TB/s.
"""
import argparse
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmda_amd import ops  # noqa: E402


def timeit(fn, iters=20, reps=5):
    """us per launch of `iters` dependent launches REPLAYED FROM A hipGraph (as tools/dbg/rp_bench.py does): eager launches from Python
    are host-bound at ~8 us each, so event-bracketed eager loops measured the launch path, not the kernel, for everything under ~10 us
    (VERDICT r05 weak #5).  The figure still includes the ~1.9-us dependent-launch boundary of a replayed graph."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            g.replay()
        e.record()
        torch.cuda.synchronize()
    return s.elapsed_time(e) / (iters * reps) * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    a = ap.parse_args()
    B = a.batch
    dev = 'cuda'
    bf = torch.bfloat16
    rows = []

    def rec(name, us, nbytes):
        rows.append((name, us, nbytes / us / 1e3))

    # MixFFN depthwise stencils per stage (hidden = 4*dim) + sep-ASPP depthwise
    for (H, C, dil) in ((128, 256, 1), (64, 512, 1), (32, 1280, 1), (16, 2048, 1), (128, 1024, 6), (128, 1024, 18)):
        n = B * H * H * C
        x = torch.randn(B, H, H, C, device=dev).to(bf)
        dy = torch.randn_like(x)
        w = torch.randn(9, C, device=dev) * 0.3
        bias = torch.randn(C, device=dev)
        dw = torch.zeros(C, 9, device=dev)
        db = torch.zeros(C, device=dev)
        tag = f'H{H} C{C} d{dil}'
        rec(f'dw fwd+gelu {tag}', timeit(lambda: ops.dwconv_fwd(x, w, bias, B, H, H, C, dil, 'gelu')), 2 * n * 2)
        rec(f'dw gelu-bwd-prep {tag}', timeit(lambda: ops.dwconv_gelu_bwd_prep(x, w, bias, dy, B, H, H, C, dil)), 3 * n * 2)
        rec(f'dw bwd-data {tag}', timeit(lambda: ops.dwconv_bwd_data(dy, w, B, H, H, C, dil)), 2 * n * 2)
        rec(f'dw bwd-weight {tag}', timeit(lambda: ops.dwconv_bwd_weight(dy, x, dw, db, B, H, H, C, dil)), 2 * n * 2)
        if dil == 1:   # the MixFFN backward's fused pass (GELU backward + depthwise weight / bias gradient): what the step runs
            rec(f'dw gelu-bwd FUSED {tag}', timeit(lambda: ops.dwconv_gelu_bwd_fused(x, w, bias, dy, dw, db, B, H, H, C, dil)), 3 * n * 2)
        del x, dy
    # LayerNorm per stage
    for (H, C) in ((128, 64), (64, 128), (32, 320), (16, 512)):
        R = B * H * H
        x = torch.randn(R, C, device=dev).to(bf)
        dy = torch.randn_like(x)
        g = torch.randn(C, device=dev)
        b = torch.randn(C, device=dev)
        y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
        dg, dbeta = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        rec(f'ln fwd R{R} C{C}', timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-6)), 2 * R * C * 2)
        rec(f'ln bwd R{R} C{C}', timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dg, dbeta)), 3 * R * C * 2)
    # decode head BatchNorm (train mode, + ReLU): joint map of G*P*B = 4 branches x 2 steps x 2 samples = 8 groups of 2 x 128 x 128
    G = 8
    Mg = 2 * 128 * 128
    for C in (256,):
        x = torch.randn(G * Mg, C, device=dev).to(bf)
        dy = torch.randn_like(x)
        y = torch.empty_like(x)
        g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        dg, dbeta = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        mean, rstd = ops.bn_train_fwd(x, g, b, y, rm, rv, Mg, C, 1e-5, 0.1, True, groups=G)
        n = G * Mg * C
        rec(f'bn fwd(+relu) {G}x{Mg} C{C}', timeit(lambda: ops.bn_train_fwd(x, g, b, y, rm, rv, Mg, C, 1e-5, 0.1, True, groups=G)), 3 * n * 2)
        rec(f'bn bwd {G}x{Mg} C{C}', timeit(lambda: ops.bn_train_bwd(dy, x, mean, rstd, g, b, dg, dbeta, Mg, C, True, groups=G)), 5 * n * 2)
    # attention softmax per stage: rows = B*heads*N, L = N/sr^2
    for (N, heads, L) in ((16384, 1, 256), (4096, 2, 256), (1024, 5, 256), (256, 8, 256)):
        R = B * heads * N
        s = torch.randn(R, L, device=dev).to(bf)
        dp = torch.randn_like(s)
        rec(f'softmax fwd R{R} L{L}', timeit(lambda: ops.softmax_fwd_(s, R, L, 0.125)), 2 * R * L * 2)
        rec(f'softmax bwd R{R} L{L}', timeit(lambda: ops.softmax_bwd_(s, dp, R, L, 0.125)), 3 * R * L * 2)
    # optimiser passes over the flat fp32 parameter store (MiT-B5 x 2 + heads = 170 M parameters in two groups of ~85 M)
    n = 85_000_000
    pp, gg, mm, vv = (torch.randn(n, device=dev) for _ in range(4))
    vv.abs_()
    mirror = torch.empty(n, device=dev, dtype=bf)
    rec(f'adamw n={n}', timeit(lambda: ops.adamw_step(pp, gg, mm, vv, 6e-5, 0.9, 0.999, 1e-8, 0.01, 3, p_bf16=mirror)), 30 * n)
    rec(f'ema n={n}', timeit(lambda: ops.ema_update(mm, pp, 0.999, mirror=mirror)), 14 * n)
    del pp, gg, mm, vv, mirror
    print(f'{"kernel":44s} {"us":>10s} {"GB/s":>9s} {"frac":>6s}')
    for name, us, gbs in rows:
        print(f'{name:44s} {us:10.1f} {gbs:9.0f} {gbs / 8000:6.3f}')


if __name__ == '__main__':
    main()
