#!/bin/bash
mkdir -p gpurun_out/loader
timeout 300 python tools/loader_bench.py > gpurun_out/loader/out.txt 2> gpurun_out/loader/err; tail -2 gpurun_out/loader/err; cat gpurun_out/loader/out.txt
