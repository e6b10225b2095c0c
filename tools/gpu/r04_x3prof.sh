#!/bin/bash
# eager kernel statistics of the split-bf16 parity mode (bench --dtype f32x3)
tag=${1:-r04x}; out=gpurun_out/$tag; mkdir -p $out
timeout 600 python bench.py --dtype f32x3 --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode > $out/bench_x3.json 2> $out/err0; cut -c1-180 $out/bench_x3.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --dtype f32x3 --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-graph > $out/prof.json 2> $out/err2
f=$(find $out/stats -name '*kernel_stats.csv' | head -1); cp $f $out/x3_eager_kernel_stats.csv; rm -rf $out/stats
python - <<PY
import csv
rows=list(csv.DictReader(open('$out/x3_eager_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms over the run', tot/1e6)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:25]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {int(r['Calls']):7d} calls avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:110]}")
PY
