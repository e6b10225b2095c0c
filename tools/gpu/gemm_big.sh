#!/bin/bash
mkdir -p gpurun_out/big
timeout 600 python tools/gemm_bench.py --big > gpurun_out/big/big.txt 2> gpurun_out/big/err; cat gpurun_out/big/big.txt
timeout 600 python tools/gemm_bench.py --batch 16 > gpurun_out/big/b16.txt 2>> gpurun_out/big/err; tail -8 gpurun_out/big/b16.txt
