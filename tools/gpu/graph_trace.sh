#!/bin/bash
# kernel trace of the GRAPH-REPLAY bench (for the gaps between iterations): gzip'd trace under gpurun_out/<tag>/
out=gpurun_out/${1:-gtrace}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity-mode > $out/bench.json 2> $out/err
t=$(find $out/kt -name '*kernel_trace.csv' | head -1); python3 - $t <<'PY'
import csv, sys, re
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    n=re.sub(r'\(.*','',r['Kernel_Name'].replace('void (anonymous namespace)::','').replace('(anonymous namespace)::',''))[:50]
    rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),n,r.get('Queue_Id','?'),r.get('Stream_Id','?')))
rows.sort()
ad=[i for i,r in enumerate(rows) if r[2].startswith('adamw')]
# the last 5 steps: for each adamw group print what surrounds it
print('launches', len(rows), 'adamw launches', len(ad))
groups=[]
for i in ad:
    if not groups or i-groups[-1][-1]>50: groups.append([i])
    else: groups[-1].append(i)
for g in groups[-5:]:
    lo=max(0,g[0]-6); hi=min(len(rows),g[-1]+14)
    t0=rows[g[0]][0]
    print('--- step boundary')
    for s,e,n,q,st in rows[lo:hi]:
        print(f'  {(s-t0)/1e3:9.1f} -> {(e-t0)/1e3:9.1f} us  q{q} s{st} {n}')
PY
rm -rf $out/kt
