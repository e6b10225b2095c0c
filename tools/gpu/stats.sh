#!/bin/bash
# interim: rocprofv3 kernel statistics of the eager launch sequence of the default bench step (per-kernel totals per step)
out=gpurun_out/${1:-r05stats}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_eager -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-parity-mode --no-graph ${BENCH_ARGS} > $out/bench_prof_eager.json 2> $out/err2
cp "$(ls -t $out/stats_eager/*/*kernel_stats.csv | head -1)" $out/eager_kernel_stats.csv
t=$(find $out/stats_eager -name '*kernel_trace.csv' | head -1)
python - $t <<'PY'
import csv, sys, re, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'\(.*', '', n)[:78]))
rows.sort()
# the LAST step: from the last ema_kernel launch to the end
marks = [i for i, r in enumerate(rows) if r[2].startswith('ema_kernel')]
it = rows[marks[-1]:]
fam = collections.Counter(); cnt = collections.Counter()
for s, e, n in it:
    fam[n] += e - s; cnt[n] += 1
print(f'last step: {len(it)} launches, kernel time {sum(fam.values()) / 1e6:.2f} ms')
for k, v in fam.most_common(45):
    print(f'  {v / 1e6:7.3f} ms  {cnt[k]:5d} x {v / cnt[k] / 1e3:8.2f} us  {k}')
PY
rm -rf $out/stats_eager
