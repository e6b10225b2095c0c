import sys, os, torch
sys.path.insert(0, '/root/repo')
from cmda_amd import ops
dev='cuda'; bf=torch.bfloat16; B=16
for (H, C) in ((128, 64), (64, 128), (32, 320), (16, 512)):
    R = B*H*H
    x = torch.randn(R, C, device=dev).to(bf); dy = torch.randn_like(x)
    g = torch.randn(C, device=dev); b = torch.randn(C, device=dev)
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    for _ in range(20):
        ops.layernorm_bwd(dy, x, g, mean, rstd, dg, db, dres=dy)
torch.cuda.synchronize()
