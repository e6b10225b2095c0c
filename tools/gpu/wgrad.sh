#!/bin/bash
mkdir -p gpurun_out/wgrad
timeout 300 python tools/dbg/wgrad3x3_dbg.py > gpurun_out/wgrad/out.txt 2>&1; cat gpurun_out/wgrad/out.txt
