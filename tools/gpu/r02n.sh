#!/bin/bash
# per-shape GEMM time table of one eager DACS step
mkdir -p gpurun_out/r02n
CMDA_BENCH_GEMM_HIST=gpurun_out/r02n/gemm_hist.txt timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02n/bench.json 2> gpurun_out/r02n/err
tail -c 300 gpurun_out/r02n/err
