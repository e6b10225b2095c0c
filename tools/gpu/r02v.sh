#!/bin/bash
mkdir -p gpurun_out/r02v
timeout 600 python tools/gemm_bench.py --big > gpurun_out/r02v/big.txt 2> gpurun_out/r02v/err; cat gpurun_out/r02v/big.txt; tail -2 gpurun_out/r02v/err
