#!/bin/bash
mkdir -p gpurun_out/dp
timeout 900 python -m pytest tests/test_dacs.py tests/test_fullsize.py tests/test_modules.py -x -q -m gpu > gpurun_out/dp/tests.log 2>&1; tail -2 gpurun_out/dp/tests.log
timeout 600 python tools/aten_census.py 2>/dev/null | head -8
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/dp/bench_$i.json 2>> gpurun_out/dp/err; python -c "
import json;d=json.loads(open('gpurun_out/dp/bench_$i.json').read().strip().splitlines()[-1]);print('run $i', d['ms_per_step'])"; done
