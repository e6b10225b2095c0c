#!/usr/bin/env python3
"""Concurrency timeline of the LAST iteration in a rocprofv3 --kernel-trace CSV of `bench.py` (graph replay): how long 0, 1, 2, ...
kernels were running at once, which kernel families ran alone, and the longest stretches with a single queue busy.
usage: timeline.py DIR"""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    n = re.sub(r'\(.*', '', n)[:48]
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n, r.get('Queue_Id', '?')))
rows.sort()
# iterations are delimited by the EMA kernel (first kernel of an iteration outside the graph)
marks = [i for i, r in enumerate(rows) if r[2].startswith('ema_kernel')]
if len(marks) >= 2:
    lo, hi = marks[-2], marks[-1]
else:
    lo, hi = 0, len(rows)
it = rows[lo:hi]
t0, t1 = it[0][0], max(r[1] for r in it)
print(f'iteration: {len(it)} kernels, {(t1 - t0) / 1e6:.2f} ms wall, kernel time {sum(r[1] - r[0] for r in it) / 1e6:.2f} ms, '
      f'queues {sorted(set(r[3] for r in it))}')
ev = []
for s, e, n, q in it:
    ev.append((s, 1, n))
    ev.append((e, -1, n))
ev.sort()
level, last = 0, t0
hist = collections.Counter()
alone = collections.Counter()
running = collections.Counter()
for t, d, n in ev:
    hist[level] += t - last
    if level == 1:
        for k, v in running.items():
            if v > 0:
                alone[k] += t - last
    last = t
    level += d
    running[n] += d
for k in sorted(hist):
    print(f'  {k} kernels running: {hist[k] / 1e6:7.2f} ms')
print('time alone on the chip, by kernel (ms):')
for k, v in alone.most_common(25):
    print(f'  {v / 1e6:7.2f}  {k}')
# gaps (idle chip) by the kernel that follows
gap_after = collections.Counter()
end = t0
for s, e, n, q in it:
    if s > end:
        gap_after[n] += s - end
    end = max(end, e)
print('idle time before kernel (ms):')
for k, v in gap_after.most_common(12):
    print(f'  {v / 1e6:7.2f}  {k}')
# stretches where ONE queue was busy (the other lane idle) for > 1 ms: kernel families inside, their time and the idle gaps
by_q = collections.defaultdict(list)
for s, e, n, q in it:
    by_q[q].append((s, e, n))
if len(by_q) >= 2:
    qs = sorted(by_q, key=lambda k: -len(by_q[k]))[:2]
    other = {qs[0]: qs[1], qs[1]: qs[0]}
    for q in qs:
        oth = sorted(by_q[other[q]])
        # idle windows of the other queue
        wins, end = [], t0
        for s, e, n in oth:
            if s - end > 1e6:
                wins.append((end, s))
            end = max(end, e)
        if t1 - end > 1e6:
            wins.append((end, t1))
        for a, b in wins:
            ks = [(s, e, n) for s, e, n in by_q[q] if s >= a and e <= b]
            if not ks:
                continue
            fam = collections.Counter()
            cnt = collections.Counter()
            for s, e, n in ks:
                fam[n] += e - s
                cnt[n] += 1
            busy = sum(fam.values())
            print(f'queue {q} alone {(a - t0) / 1e6:.2f} -> {(b - t0) / 1e6:.2f} ms: {len(ks)} kernels, busy {busy / 1e6:.2f} ms of {(b - a) / 1e6:.2f}')
            for k, v in fam.most_common(30):
                print(f'    {v / 1e6:7.3f} ms  {cnt[k]:4d} x {v / cnt[k] / 1e3:8.1f} us  {k}')
