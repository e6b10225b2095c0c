#!/bin/bash
mkdir -p gpurun_out/lastcheck
timeout 1200 python -m pytest tests/test_dacs.py tests/test_kernels.py -x -q -m gpu > gpurun_out/lastcheck/tests.log 2>&1; tail -3 gpurun_out/lastcheck/tests.log
timeout 600 python bench.py > gpurun_out/lastcheck/bench.json 2> gpurun_out/lastcheck/err; cut -c1-230 gpurun_out/lastcheck/bench.json
