#!/bin/bash
mkdir -p gpurun_out/ce
timeout 300 python tools/dbg/ce_dbg.py > gpurun_out/ce/out.txt 2>&1; cat gpurun_out/ce/out.txt
timeout 1200 python -m pytest tests/test_kernels.py tests/test_fullsize.py tests/test_dacs.py tests/test_modules.py -x -q -m gpu > gpurun_out/ce/tests.log 2>&1; tail -2 gpurun_out/ce/tests.log
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ce/bench_$i.json 2>> gpurun_out/ce/err; python -c "
import json;d=json.loads(open('gpurun_out/ce/bench_$i.json').read().strip().splitlines()[-1]);print('run $i', d['ms_per_step'])"; done
