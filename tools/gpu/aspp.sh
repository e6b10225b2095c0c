#!/bin/bash
mkdir -p gpurun_out/aspp
timeout 1200 python -m pytest tests/test_dacs.py tests/test_modules.py tests/test_parallel.py -x -q -m gpu > gpurun_out/aspp/tests.log 2>&1; tail -3 gpurun_out/aspp/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/aspp/bench.json 2> gpurun_out/aspp/err; cut -c1-230 gpurun_out/aspp/bench.json
timeout 600 python tools/lanes_timeline.py > gpurun_out/aspp/timeline.txt 2>> gpurun_out/aspp/err; cat gpurun_out/aspp/timeline.txt
