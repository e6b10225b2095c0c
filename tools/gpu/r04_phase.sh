#!/bin/bash
out=gpurun_out/${1:-r04g}; mkdir -p $out
RP_SHORT=1 python tools/dbg/rp_bench.py > $out/floor.txt 2>&1; cat $out/floor.txt
CMDA_HIP_LIB=build/libcmda_hip_timing.so python tools/gemm_phase.py > $out/gemm_phase.txt 2>&1; cat $out/gemm_phase.txt
python - <<'PY' > $out/wg_ab.txt 2>&1
import sys, os, torch
sys.path.insert(0, os.getcwd())
sys.argv = ['x']
from cmda_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
B, H, W, Ci, Co, k = 16, 128, 128, 1024, 256, 3
x = torch.randn(B * H * W, Ci, device=dev).bfloat16(); dy = torch.randn(B * H * W, Co, device=dev).bfloat16()
M, K = B * H * W, k * k * Ci
dw = torch.zeros(Co, K, device=dev)
for hint, name in ((0, 'default (64,2) row-fast'), (4096, '(32,4) row-fast'), (2048, 'general path')):
    ops.GEMM_TILE_HINT = hint
    t = timeit(lambda: ops.gemm(ops.plain_view(dy, M, Co), ops.conv_view(x, B, H, W, Ci, k, k, 1, 1), dw, Co, K, M, a_kstrided=True, b_kstrided=True, dtype=1, atomic=True, splits=0))
    print(f'bottleneck wgrad {name}: {t:.1f} us  {2.0 * M * Co * K / t / 1e6:.0f} TFLOP/s')
PY
cat $out/wg_ab.txt
