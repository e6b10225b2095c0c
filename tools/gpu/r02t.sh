#!/bin/bash
mkdir -p gpurun_out/r02t
timeout 900 python -m pytest tests/test_dacs.py -x -q -m gpu > gpurun_out/r02t/tests.log 2>&1; tail -3 gpurun_out/r02t/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02t/bench.json 2> gpurun_out/r02t/err_bench; cut -c1-250 gpurun_out/r02t/bench.json
