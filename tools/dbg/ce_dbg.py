"""CE forward / backward / pseudo-label at the step's shape (2 x 19 x 128 x 128 -> 512 x 512)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmda_amd import ops
dev = torch.device('cuda:0')


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


B, h, w, H, W, nc = 2, 128, 128, 512, 512, 19
logits = torch.randn(B, h, w, nc, device=dev)
label = torch.randint(0, 19, (B, H, W), device=dev)
wgt = torch.rand(B, H, W, device=dev)
acc = torch.zeros(2, device=dev)
t_f = timeit(lambda: ops.ce_upsample_fwd(logits, label, wgt, H, W, 255, acc=acc))
_, lse = ops.ce_upsample_fwd(logits, label, wgt, H, W, 255)
t_b = timeit(lambda: ops.ce_upsample_bwd(logits, label, wgt, lse, None, 1.0, H, W, 255))
t_p = timeit(lambda: ops.pseudo_label(logits, H, W, 0.968, want_prob=False))
print(f'ce fwd {t_f:.1f} us, ce bwd {t_b:.1f} us, pseudo-label {t_p:.1f} us (B=2, 128x128 -> 512x512, 19 classes)')
