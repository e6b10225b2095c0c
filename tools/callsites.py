#!/usr/bin/env python3
"""Launch census of one eager DACS iteration at the bench's own configuration: C-ABI calls per entry point with their Python call
sites (where do the ~5.5 k launches of a step come from?).  usage: python tools/callsites.py [entry_point ...]"""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cmda_amd import ops, optim  # noqa: E402


def main():
    from cmda_amd import runtime as rt
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
    cnt, sites = collections.Counter(), collections.defaultdict(collections.Counter)
    orig = ops.call

    def counted(name, *a):
        cnt[name] += 1
        st = traceback.extract_stack(limit=7)
        site = ' < '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in reversed(st[:-1]) if 'cmda_amd' in f.filename)[:120]
        sites[name][site] += 1
        return orig(name, *a)
    ops.call = counted
    opt.zero_grad()
    dacs(**batch)
    opt.step(1.0)
    torch.cuda.synchronize()
    ops.call = orig
    print('total', sum(cnt.values()))
    for k, v in cnt.most_common():
        print(f'{v:6d} {k}')
    for name in (sys.argv[1:] or ['cmda_permute4', 'cmda_cast_clear', 'cmda_axpby', 'cmda_sample_scale']):
        print('---', name)
        for s, v in sites[name].most_common(14):
            print(f'   {v:5d} {s}')


if __name__ == '__main__':
    main()
