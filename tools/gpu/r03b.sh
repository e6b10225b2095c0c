#!/bin/bash
mkdir -p gpurun_out/r03b
for lanes in enc enc,wgrad enc; do
  CMDA_BENCH_LANES=$lanes timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03b/bench_$lanes.json 2> gpurun_out/r03b/err_$lanes
  python -c "
import json;d=json.loads(open('gpurun_out/r03b/bench_$lanes.json').read().strip().splitlines()[-1]);print('$lanes', d['ms_per_step'], d['losses'])"
done
