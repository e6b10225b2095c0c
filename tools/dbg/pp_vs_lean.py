import os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ.pop('RP_SHORT', None)
sys.argv=['x']
import importlib.util
spec = importlib.util.spec_from_file_location('rp', 'tools/dbg/rp_bench.py')
src = open('tools/dbg/rp_bench.py').read().split("_t = torch.zeros(256, device=dev)")[0]
exec(src)
for M, N, K in ((131072, 256, 64), (65536, 256, 64), (32768, 512, 128), (16384, 512, 128), (131072, 64, 256), (32768, 128, 512), (65536, 1280, 320), (16384, 1280, 320)):
    t = [nt(M, N, K, h) for h in (0, 3, 2, 1)]
    fl = 2.0 * M * N * K
    print(f'{M:7d} x {N:5d} x {K:5d}: heuristics {t[0]:7.1f} us | 64x64 {t[1]:7.1f} | 128x64 {t[2]:7.1f} | 128x128 {t[3]:7.1f}   (out {M*N*2/1e6:.0f} MB)')
