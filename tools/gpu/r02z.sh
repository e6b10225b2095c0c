#!/bin/bash
mkdir -p gpurun_out/r02z
timeout 300 python tools/dbg/srconv_dbg.py > gpurun_out/r02z/sr.txt 2>&1; cat gpurun_out/r02z/sr.txt
timeout 1500 python -m pytest tests/test_gemm.py tests/test_modules.py tests/test_dacs.py tests/test_parallel.py tests/test_fullsize.py -x -q -m gpu > gpurun_out/r02z/tests.log 2>&1; tail -3 gpurun_out/r02z/tests.log
CMDA_BENCH_GEMM_HIST=gpurun_out/r02z/gemm_hist.txt timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02z/bench.json 2> gpurun_out/r02z/err_bench; cut -c1-250 gpurun_out/r02z/bench.json
