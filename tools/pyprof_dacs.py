"""cProfile of a few full DACS iterations (host side): where the ~14 us per kernel launch of Python goes."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cmda_amd.runtime as rt  # noqa: E402
from cmda_amd import optim  # noqa: E402

dev = torch.device('cuda:0')
rt.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
dacs = bench.build_dacs(dev)
opt = optim.FlatAdamW(dacs.model, lr=6e-5, weight_decay=0.01,
                      custom_keys=dict(head=dict(lr_mult=10.0), pos_block=dict(decay_mult=0.0), norm=dict(decay_mult=0.0)))
dacs.attach_flat_store(opt)
B, S = 2, 512
g = torch.Generator().manual_seed(1)
lab = torch.randint(0, 19, (B, 1, S // 32, S // 32), generator=g).repeat_interleave(32, 2).repeat_interleave(32, 3)
r = lambda: torch.randn(B, 3, S, S, generator=g).clamp(-1, 1).to(dev)  # noqa: E731
batch = dict(source=dict(image=r(), img_time_res=r(), img_self_res=r(), label=lab.to(dev)),
             target=dict(warp_image=r(), events_vg=r(), warp_img_self_res=r()))


def step():
    opt.zero_grad()
    dacs(**batch)
    opt.step(1.0)


for _ in range(2):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
