#!/bin/bash
mkdir -p gpurun_out/drain
timeout 900 python -m pytest tests/test_dacs.py tests/test_fullsize.py -x -q -m gpu > gpurun_out/drain/tests.log 2>&1; tail -2 gpurun_out/drain/tests.log
timeout 600 python tools/lanes_timeline.py > gpurun_out/drain/timeline.txt 2> gpurun_out/drain/err; tail -4 gpurun_out/drain/timeline.txt
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/drain/bench_$i.json 2>> gpurun_out/drain/err; python -c "
import json;d=json.loads(open('gpurun_out/drain/bench_$i.json').read().strip().splitlines()[-1]);print('run $i', d['ms_per_step'])"; done
