#!/bin/bash
mkdir -p gpurun_out/r02q
timeout 600 python tools/gemm_sweep.py > gpurun_out/r02q/sweep.txt 2> gpurun_out/r02q/err
tail -3 gpurun_out/r02q/err
timeout 300 python -m pytest tests/test_kernels.py -x -q -m gpu -k "ce_" > gpurun_out/r02q/ce.log 2>&1; tail -2 gpurun_out/r02q/ce.log
