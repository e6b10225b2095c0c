#!/bin/bash
mkdir -p gpurun_out/hbm
timeout 300 python tools/hbm_bench.py --batch 8 > gpurun_out/hbm/b8.txt 2> gpurun_out/hbm/err; head -18 gpurun_out/hbm/b8.txt
