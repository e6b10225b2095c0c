import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmda_amd import ops
dev = torch.device('cuda:0')
for hint, M, N in [(3, 600, 1100), (3, 520, 1024), (2, 1100, 1060), (4, 1030, 2100)]:
    K = 64
    torch.manual_seed(hint * 100 + M)
    a, b = torch.randn(M, K).to(torch.bfloat16), torch.randn(N, K).to(torch.bfloat16)
    ref = a.float() @ b.float().t()
    ad, bd = a.to(dev), b.to(dev)
    for h in (hint, hint | 256):
        ops.GEMM_TILE_HINT = h
        for rep in range(3):
            out = torch.full((M, N), float('nan'), dtype=torch.bfloat16, device=dev)
            ops.gemm(ops.plain_view(ad, M, K), ops.plain_view(bd, N, K), out, M, N, K, dtype=1)
            torch.cuda.synchronize()
            d = (out.float().cpu() - ref).abs()
            bad = (d > 0.6) | d.isnan()
            bm = {3: 64, 2: 128, 4: 256}[hint]; bn = {3: 64, 2: 64, 4: 256}[hint]
            tiles = sorted(set((int(i) // bm, int(j) // bn) for i, j in bad.nonzero()[:20000].tolist()))
            print(f'hint {h} M {M} N {N} rep {rep}: max err {d.nan_to_num(9999).max().item():.3f}, bad elems {int(bad.sum())}, bad tiles {tiles[:12]}{"..." if len(tiles) > 12 else ""}')
ops.GEMM_TILE_HINT = 0
