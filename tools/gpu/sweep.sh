#!/bin/bash
mkdir -p gpurun_out/sweep
timeout 600 python tools/gemm_sweep.py > gpurun_out/sweep/sweep.txt 2> gpurun_out/sweep/err; tail -2 gpurun_out/sweep/err; cat gpurun_out/sweep/sweep.txt
