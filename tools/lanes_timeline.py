#!/usr/bin/env python3
"""Gantt of one replayed DACS iteration at the bench's configuration: per hipGraph segment its stream, start and end (events
recorded around each segment's replay), i.e. where each of the two lanes is busy and where it waits for the other."""
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
    opt.overlap = os.environ.get('CMDA_OPT_OVERLAP', '1') != '0'   # as bench.py: the step boundary on the optimizer's own stream
    dacs.attach_flat_store(opt)
    batch = bench.synthetic_pairs(2, 512, 100, dev)
    dacs.enable_graph(warmup_iters=2)
    for _ in range(6):
        opt.zero_grad()
        dacs(**batch)
        opt.step(1.0)
    torch.cuda.synchronize()
    seg = dacs._graph['graph']
    tl = []
    orig = seg.replay
    seg.replay = lambda: orig(timeline=tl)
    marks = []
    for _ in range(3):                      # three timed iterations: the first one's segments are printed, the gaps between them too
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(('iteration start (host reaches zero_grad)', e))
        opt.zero_grad()
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(('zero_grad queued', e))
        dacs(**batch)
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(('graph replay queued', e))
        opt.step(1.0)
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(('AdamW queued', e))
    torch.cuda.synchronize()
    seg.replay = orig
    nseg = len(tl) // 3
    t0 = tl[0][2]
    print('events on the main stream and segment boundaries of three consecutive iterations (ms from the first segment):')
    for name, e in marks:
        print(f'  {t0.elapsed_time(e):9.3f}  {name}')
    for k in range(3):
        a = min(t0.elapsed_time(r[2]) for r in tl[k * nseg:(k + 1) * nseg])
        b = max(t0.elapsed_time(r[3]) for r in tl[k * nseg:(k + 1) * nseg])
        print(f'  iteration {k}: first segment starts {a:9.3f}, last segment ends {b:9.3f}  (span {b - a:.3f})')
    tl = tl[:nseg]
    streams = {}
    rows = []
    for i, sid, e0, e1 in tl:
        lane = streams.setdefault(sid, len(streams))
        rows.append((t0.elapsed_time(e0), t0.elapsed_time(e1), lane, i))
    print(f'{len(rows)} segments on {len(streams)} streams; iteration span {max(r[1] for r in rows):.2f} ms')
    busy = [0.0] * len(streams)
    for a, b, lane, i in rows:
        busy[lane] += b - a
        print(f'  seg {i:3d} lane {lane}  {a:8.2f} -> {b:8.2f}  ({b - a:6.2f} ms)')
    print('busy per lane (ms):', [round(v, 2) for v in busy])


if __name__ == '__main__':
    main()
