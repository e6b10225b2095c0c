#!/usr/bin/env python3
"""torch (aten) kernels launched inside one eager DACS iteration at the bench's configuration, by Python call site: everything here is
a launch the step pays for outside the C ABI (fills, RNG, small elementwise ops)."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cmda_amd import optim, runtime as rt  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    rt.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(1234)
    dacs = bench.build_dacs(dev)
    opt = optim.FlatAdamW(dacs.model, lr=6e-5, weight_decay=0.01, custom_keys=bench.CUSTOM_KEYS)
    dacs.attach_flat_store(opt)
    batch = bench.synthetic_pairs(2, 512, 100, dev)
    for _ in range(2):
        opt.zero_grad()
        dacs(**batch)
        opt.step(1.0)
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        opt.zero_grad()
        dacs(**batch)
        opt.step(1.0)
        torch.cuda.synchronize()
    cnt = collections.Counter()
    for e in prof.events():
        if e.name.startswith('aten::') and len(getattr(e, 'kernels', ())) > 0:
            st = [s for s in e.stack if 'cmda_amd' in s or 'bench.py' in s][:2]
            cnt[(e.name, ' <- '.join(s.split('/')[-1] for s in st))] += 1
    tot = sum(cnt.values())
    print('aten ops that launched kernels:', tot)
    for (n, s), v in cnt.most_common(40):
        print(f'{v:5d} {n:28s} {s}')


if __name__ == '__main__':
    main()
