"""tuning experiment: block shape of the depthwise weight-gradient kernel at the decode head's shapes (CMDA_DW_CQ / CMDA_DW_K)"""
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
for (B, H, C, dil) in ((16, 128, 1024, 6), (16, 128, 1024, 18), (4, 128, 256, 1), (4, 32, 1280, 1)):
    x = torch.randn(B, H, H, C, device='cuda').to(bf); dy = torch.randn_like(x)
    dw = torch.zeros(C, 9, device='cuda'); db = torch.zeros(C, device='cuda')
    out = []
    for cq, k in (("64", None), ("16", None), ("16", "4"), ("16", "8"), ("16", "32"), ("64", "8"), ("64", "32"), ("64", "64")):
        os.environ["CMDA_DW_CQ"] = cq
        if k: os.environ['CMDA_DW_K'] = k
        else: os.environ.pop('CMDA_DW_K', None)
        out.append(f'cq{cq}/k{k}: {timeit(lambda: ops.dwconv_bwd_weight(dy, x, dw, db, B, H, H, C, dil)):6.1f}')
    print(f'B{B} H{H} C{C} d{dil}: ' + '  '.join(out))
