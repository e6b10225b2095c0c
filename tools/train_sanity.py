#!/usr/bin/env python3
"""Sanity of the whole training loop at the bench's configuration: N graph-replayed UDA iterations on a few fixed synthetic batches, poly-warm
learning rate -- the source / mixed losses must fall and stay finite."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cmda_amd import optim, runtime as rt  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    dev = torch.device('cuda:0')
    rt.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(1234)
    dacs = bench.build_dacs(dev)
    opt = optim.FlatAdamW(dacs.model, lr=6e-5, weight_decay=0.01, custom_keys=bench.CUSTOM_KEYS)
    opt.overlap = os.environ.get('CMDA_OPT_OVERLAP', '1') != '0'   # as bench.py: the step boundary on the optimizer's own stream
    dacs.attach_flat_store(opt)
    batches = [bench.synthetic_pairs(2, 512, 100 + i, dev) for i in range(4)]
    dacs.enable_graph(warmup_iters=2)
    for it in range(n):
        opt.zero_grad()
        lv = dacs(**batches[it % len(batches)])
        opt.step(optim.poly_warm_scale(it + 1400))   # near the end of the warm-up: a learning rate that moves the weights
        if it % 10 == 0 or it == n - 1:
            vals = {k: float(v) for k, v in lv.items() if 'loss' in k}
            print(it, {k: round(v, 4) for k, v in vals.items()}, flush=True)
            assert all(v == v and abs(v) < 1e4 for v in vals.values()), 'loss is not finite'
    print('ok')


if __name__ == '__main__':
    main()
