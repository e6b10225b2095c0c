#!/usr/bin/env python3
"""Per-phase cycle counts of the fused MixFFN kernel (build: `make mftiming`, run with CMDA_HIP_LIB=build/libcmda_hip_mftiming.so):
s_memtime deltas accumulated by lane 0 of wave 0 of workgroup 0 over the whole kernel."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmda_amd import _lib as L, ops  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

dev = torch.device('cuda:0')
B, save = int(sys.argv[1]) if len(sys.argv) > 1 else 4, (sys.argv[2] == 'save') if len(sys.argv) > 2 else False
H = W = 32
C, hidden = 320, 1280
M = B * H * W
x = torch.randn(M, C, device=dev)
args = (torch.randn(C, device=dev), torch.randn(C, device=dev), 1e-6, (torch.randn(hidden, C, device=dev) * C ** -0.5).bfloat16(),
        torch.randn(hidden, device=dev) * 0.1, torch.randn(9, hidden, device=dev) * 0.3, torch.randn(hidden, device=dev) * 0.1,
        (torch.randn(C, hidden, device=dev) * hidden ** -0.5).bfloat16(), torch.randn(C, device=dev) * 0.1, None, B, H, W)
for _ in range(3):
    ops.mixffn_fwd(x, *args, save=save)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
assert L.lib().cmda_debug_mixffn_stamps(buf) == 0
st = list(buf)
names = ['prologue (DMA issue + LayerNorm)', 'A fragments', 'fc1 MFMA steps', 'h epilogue + barrier', 'stencil + barrier', 'fc2 MFMA steps',
         'epilogue', 'TOTAL', 'wait W1 (top barrier)', 'wait W2 (+ activation hand-off)', 'h epilogue compute + LDS stores', 'prologue: address setup + load issue', 'prologue: statistics (waits for the rows)', 'prologue: normalise + LDS stores', 'prologue: first barrier of a pass', 'x']
print(f'B = {B}, save = {save}: s_memtime ticks (100 MHz = 10 ns each unless the counter is the shader clock)')
for n, v in zip(names, st):
    print(f'  {n:36s} {v:8d}')
