#!/bin/bash
mkdir -p gpurun_out/headgemm
timeout 300 python tools/dbg/head_gemm_dbg.py > gpurun_out/headgemm/out.txt 2>&1; cat gpurun_out/headgemm/out.txt
