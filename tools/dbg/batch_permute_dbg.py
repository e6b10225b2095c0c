import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cmda_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
shapes = [(64, 3, 7, 7), (64, 64, 8, 8), (128, 64, 3, 3), (128, 128, 4, 4), (320, 128, 3, 3), (320, 320, 2, 2), (512, 320, 3, 3), (256, 1024, 3, 3), (256, 1, 3, 3), (1024, 1, 3, 3), (19, 256, 1, 1)] * 3
for trial in range(3):
    srcs = [torch.randn(s, device=dev) for s in shapes]
    GUARD = 257
    specs = []
    for s_, p in zip(shapes, srcs):
        Co, Ci, KH, KW = s_
        specs.append((p, (Co, Ci, KH, KW), (0, 2, 3, 1), 0, torch.float32))
        specs.append((p, (Co, Ci, KH, KW), (1, 2, 3, 0), 0b1100, torch.bfloat16))
    sizes = [p.numel() for p, *_ in specs]
    arena32 = torch.full((sum(sizes) + GUARD * (len(sizes) + 1),), 777.0, device=dev)
    arena16 = torch.full((sum(sizes) + GUARD * (len(sizes) + 1),), 777.0, device=dev, dtype=torch.bfloat16)
    desc = np.zeros(len(specs), dtype=[('src', '<u8'), ('dst', '<u8'), ('d', '<i4', 4), ('p', '<i4', 4), ('flip', '<i4'), ('bf16', '<i4'), ('total', '<i8')])
    blocks, dsts, off = [], [], GUARD
    for t, (p, d, pm, flip, dt) in enumerate(specs):
        arena = arena32 if dt == torch.float32 else arena16
        dst = arena[off:off + p.numel()]
        dsts.append(dst)
        desc[t] = (p.data_ptr(), dst.data_ptr(), d, pm, flip, int(dt == torch.bfloat16), p.numel())
        blocks += [(t, c) for c in range((p.numel() + 1023) // 1024)]
        off += p.numel() + GUARD
    dd = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).to(dev)
    bb = torch.tensor(blocks, dtype=torch.int32).to(dev)
    ops.permute4_batch(dd, bb, len(blocks))
    torch.cuda.synchronize()
    bad = 0
    for (p, d, pm, flip, dt), dst in zip(specs, dsts):
        ref = torch.empty(p.numel(), dtype=dt, device=dev)
        ops.permute4(p, ref, d, pm, flipmask=flip)
        if not torch.equal(ref, dst): bad += 1
    used32 = torch.zeros_like(arena32, dtype=torch.bool); used16 = torch.zeros_like(arena16, dtype=torch.bool)
    off = GUARD
    for (p, d, pm, flip, dt) in specs:
        (used32 if dt == torch.float32 else used16)[off:off + p.numel()] = True
        off += p.numel() + GUARD
    g32 = (arena32[~used32] != 777.0).sum().item(); g16 = (arena16[~used16].float() != 777.0).sum().item()
    print('trial', trial, 'tensors', len(specs), 'bad', bad, 'guard violations', g32, g16, 'blocks', len(blocks))
