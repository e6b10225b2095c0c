#!/bin/bash
mkdir -p gpurun_out/r03a
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03a/bench_loader.json 2> gpurun_out/r03a/err1; tail -3 gpurun_out/r03a/err1; cut -c1-200 gpurun_out/r03a/bench_loader.json
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --data direct > gpurun_out/r03a/bench_direct.json 2> gpurun_out/r03a/err2; cut -c1-200 gpurun_out/r03a/bench_direct.json
python - <<'PY'
import json
for f in ('loader','direct'):
    d=json.loads(open(f'gpurun_out/r03a/bench_{f}.json').read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['losses'])
PY
