mkdir -p gpurun_out/r02e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02e/prof -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02e/bench_prof.json 2> gpurun_out/r02e/bench_prof.err
tail -2 gpurun_out/r02e/bench_prof.err; cut -c1-300 gpurun_out/r02e/bench_prof.json
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/r02e/prof/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
# last 30% of the trace = replays
n = len(rows)
seg = rows[int(n * 0.75):]
q = collections.Counter(r['Queue_Id'] for r in seg)
print('queues in the tail segment:', q)
wall = int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
# union of intervals
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in seg)
u = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: u += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
u += ce - cs
print(f'tail segment: {len(seg)} kernels, wall {wall/1e6:.1f} ms, sum of durations {busy/1e6:.1f} ms, union {u/1e6:.1f} ms, idle {(wall-u)/1e6:.1f} ms')
gaps = [int(seg[i+1]['Start_Timestamp']) - int(seg[i]['End_Timestamp']) for i in range(len(seg)-1)]
import statistics
print('median gap between consecutive kernels (by start order) ns:', statistics.median(gaps))
PY
# keep the trace small for the merge-back
rm -f gpurun_out/r02e/prof/*/*_kernel_trace.csv
