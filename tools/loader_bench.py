#!/usr/bin/env python3
"""Time of one training batch (2 source + 2 target samples) through the on-device loader pipeline, synthetic raw streams generated
on the host (that generation is timed separately: it stands in for file decoding, which is outside the hot path)."""
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cmda_amd import datasets  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    S = 512
    cfg = dict(type='UDADataset',
               source=dict(type='CityscapesICDataset', image_resize_size=(2 * S, S), image_crop_size=(S, S),
                           outputs={'image', 'label', 'img_time_res', 'img_self_res'}, isr_parms=dict(bench.ISR_PARMS),
                           shift_type='random', synthetic_length=64, device=dev),
               target=dict(type='DSECDataset', crop_size=(400, 400), after_crop_resize_size=(S, S), events_bins=1,
                           isr_parms=dict(bench.ISR_PARMS), outputs={'warp_image', 'events_vg', 'warp_img_self_res'},
                           shift_type='random', synthetic_length=64, synthetic_events=500000, device=dev))
    ds = datasets.build_dataset(cfg)
    random.seed(0)
    # host-side synthetic raw data (stand-in for decoding), timed apart
    t0 = time.perf_counter()
    for i in range(4):
        ds.source.raw(i)
        ds.target.raw(i)
    t_raw = (time.perf_counter() - t0) / 2
    for _ in range(2):
        ds.get_batch([0, 1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for k in range(n):
        ds.get_batch([2 * k, 2 * k + 1])
    torch.cuda.synchronize()
    t_batch = (time.perf_counter() - t0) / n
    if os.environ.get('LOADER_PROFILE'):
        import cProfile, pstats
        import functools
        ds.source.raw = functools.lru_cache(None)(ds.source.raw)
        ds.target.raw = functools.lru_cache(None)(ds.target.raw)
        ds.get_batch([20, 21]), ds.get_batch([22, 23])   # fills the raw-data caches: only the pipeline is profiled below
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        for k in range(2):
            ds.get_batch([20 + 2 * k, 21 + 2 * k])
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
    print(f'get_batch (2 + 2 samples, incl. host raw-data synthesis and H2D copies): {t_batch * 1e3:.1f} ms per batch; '
          f'host raw-data synthesis alone: {t_raw * 1e3:.1f} ms per batch')


if __name__ == '__main__':
    main()
