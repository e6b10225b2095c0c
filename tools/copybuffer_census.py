#!/usr/bin/env python3
"""Where do the `__amd_rocclr_copyBuffer` launches of a step come from?  
Reads a rocprofv3 --kernel-trace CSV directory and prints, for the copy kernels, the histogram of the kernel that FOLLOWS
each one on the same queue (the consumer of the copy) and of the one that precedes it, plus the copies per iteration
(iterations delimited by the EMA kernel).
usage: copybuffer_census.py DIR"""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    n = re.sub(r'\(.*', '', n)[:70]
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n, r.get('Queue_Id', '?'),
                 int(r['Grid_Size_X']), int(r['Workgroup_Size_X'])))
rows.sort()
by_q = collections.defaultdict(list)
for r in rows:
    by_q[r[3]].append(r)
nxt, prv, grids = collections.Counter(), collections.Counter(), collections.Counter()
total = 0
for q, rs in by_q.items():
    for i, r in enumerate(rs):
        if 'copyBuffer' not in r[2]:
            continue
        total += 1
        grids[(r[4], r[5])] += 1
        j = i + 1
        while j < len(rs) and 'copyBuffer' in rs[j][2]:
            j += 1
        nxt[rs[j][2] if j < len(rs) else '<end>'] += 1
        j = i - 1
        while j >= 0 and 'copyBuffer' in rs[j][2]:
            j -= 1
        prv[rs[j][2] if j >= 0 else '<start>'] += 1
marks = [i for i, r in enumerate(rows) if r[2].startswith('ema_kernel')]
print(f'{total} copyBuffer launches in {len(rows)} kernels on queues {sorted(by_q)}')
for a, b in zip(marks, marks[1:]):
    it = rows[a:b]
    print(f'  iteration of {len(it)} kernels: {sum("copyBuffer" in r[2] for r in it)} copyBuffer')
print('grid x block of the copies:', grids.most_common(8))
print('--- kernel FOLLOWING a copy (same queue)')
for k, v in nxt.most_common(25):
    print(f'{v:7d}  {k}')
print('--- kernel PRECEDING a copy (same queue)')
for k, v in prv.most_common(25):
    print(f'{v:7d}  {k}')
