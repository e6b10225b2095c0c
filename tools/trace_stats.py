"""Average kernel durations from a rocprofv3 --kernel-trace CSV directory.
usage: trace_stats.py DIR [min_calls] [--runs]
default: one line per (kernel, grid); --runs: one line per run of consecutive identical launches, in launch order (maps a
micro-benchmark's shape list onto the trace)."""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
args = [a for a in sys.argv[2:] if not a.startswith('--')]
minc = int(args[0]) if args else 10
runs = '--runs' in sys.argv
rows = []
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].replace('void (anonymous namespace)::', '')
    m = re.match(r'([\w:]+(<[^(]*>)?)', n)
    n = (m.group(1) if m else n)[:60]
    k = (n, int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), r['Grid_Size_Y'], r['Grid_Size_Z'])
    rows.append((int(r['Start_Timestamp']), k, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
rows.sort()


def show(k, n, tot):
    if n >= minc:
        print(f'{k[0]:62s} grid {k[1]:>6d} {k[2]:>5s} {k[3]:>4s} n={n:5d} avg={tot / n:9.1f} us')


if runs:
    cur, n, tot = None, 0, 0.0
    for _, k, d in rows:
        if k != cur:
            if cur:
                show(cur, n, tot)
            cur, n, tot = k, 0, 0.0
        n += 1
        tot += d
    show(cur, n, tot)
else:
    agg = collections.OrderedDict()
    for _, k, d in rows:
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += d
    for k, v in agg.items():
        show(k, v[0], v[1])
