"""tuning experiment: channel quads per block of the dilated depthwise kernel (CMDA_DW_CQ) at the decode head's shape"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmda_amd import ops

def timeit(f, iters=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

bf = torch.bfloat16
B, H, C = 16, 128, 1024
x = torch.randn(B, H, H, C, device='cuda').to(bf); dy = torch.randn_like(x); acc = torch.zeros_like(x)
w = torch.randn(9, C, device='cuda') * 0.3
y = torch.empty_like(x)
t0 = timeit(lambda: y.copy_(x))
print(f'copy_ 537 MB: {t0:.1f} us')
for dil in (6, 12, 18):
    out = []
    for cq in ('64', '128', '256'):
        os.environ['CMDA_DW_CQ'] = cq
        tf = timeit(lambda: ops.dwconv_fwd(x, w, None, B, H, H, C, dil, None))
        td = timeit(lambda: ops.dwconv_bwd_data(dy, w, B, H, H, C, dil))
        ta = timeit(lambda: ops.dwconv_bwd_data(dy, w, B, H, H, C, dil, out=acc, accumulate=True))
        out.append(f'cq{cq}: fwd {tf:6.1f} data {td:6.1f} data+acc {ta:6.1f}')
    print(f'd{dil}: ' + ' | '.join(out))
