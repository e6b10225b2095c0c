#!/bin/bash
mkdir -p gpurun_out/tileab
timeout 600 python tools/gemm_sweep.py > gpurun_out/tileab/sweep.txt 2> gpurun_out/tileab/err; grep " 1 1 1 " gpurun_out/tileab/sweep.txt | cut -c1-60
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/tileab/bench_$i.json 2> gpurun_out/tileab/err_$i
python -c "
import json;d=json.loads(open('gpurun_out/tileab/bench_$i.json').read().strip().splitlines()[-1]);print('run $i', d['ms_per_step'], d['roofline']['gemm_ms_per_step'])"
done
timeout 900 python -m pytest tests/test_gemm.py tests/test_dacs.py -x -q -m gpu > gpurun_out/tileab/tests.log 2>&1; tail -2 gpurun_out/tileab/tests.log
