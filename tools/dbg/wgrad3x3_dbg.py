"""decode-head 3x3 bottleneck weight gradient (M 256 x N 9216 x K 262144): im2col view vs a materialised plain operand of the same
shape -- how much of the kernel is im2col address generation?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmda_amd import ops
dev = torch.device('cuda:0')
bf = torch.bfloat16


def timeit(fn, iters=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


B, H, W, Ci, Co = 16, 128, 128, 1024, 256
M, K = B * H * W, 9 * Ci
x = torch.randn(M, Ci, device=dev).to(bf)
dy = torch.randn(M, Co, device=dev).to(bf)
dw = torch.zeros(Co, K, device=dev)
t_conv = timeit(lambda: ops.gemm(ops.plain_view(dy, M, Co), ops.conv_view(x, B, H, W, Ci, 3, 3, 1, 1), dw, Co, K, M, a_kstrided=True, b_kstrided=True,
                                 dtype=1, atomic=True, splits=0))
col = torch.randn(M, K, device=dev).to(bf)
t_plain = timeit(lambda: ops.gemm(ops.plain_view(dy, M, Co), ops.plain_view(col, M, K), dw, Co, K, M, a_kstrided=True, b_kstrided=True, dtype=1,
                                  atomic=True, splits=0))
fl = 2.0 * M * Co * K
print(f'wgrad 256 x 9216 x {M}: im2col view {t_conv:8.1f} us ({fl / t_conv / 1e6:5.0f} TF), plain operand {t_plain:8.1f} us ({fl / t_plain / 1e6:5.0f} TF)')
for hint in (1, 2, 3, 4):
    ops.GEMM_TILE_HINT = hint
    t = timeit(lambda: ops.gemm(ops.plain_view(dy, M, Co), ops.conv_view(x, B, H, W, Ci, 3, 3, 1, 1), dw, Co, K, M, a_kstrided=True, b_kstrided=True,
                                dtype=1, atomic=True, splits=0))
    print(f'  forced tile hint {hint}: {t:8.1f} us ({fl / t / 1e6:5.0f} TF)')
ops.GEMM_TILE_HINT = 0
