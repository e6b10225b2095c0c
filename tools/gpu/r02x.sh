#!/bin/bash
mkdir -p gpurun_out/r02x
timeout 900 python -m pytest tests/test_modules.py tests/test_gemm.py tests/test_dacs.py -x -q -m gpu > gpurun_out/r02x/tests.log 2>&1; tail -3 gpurun_out/r02x/tests.log
CMDA_BENCH_GEMM_HIST=gpurun_out/r02x/gemm_hist.txt timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02x/bench.json 2> gpurun_out/r02x/err_bench; cut -c1-250 gpurun_out/r02x/bench.json
head -12 gpurun_out/r02x/gemm_hist.txt
