#!/bin/bash
mkdir -p gpurun_out/r02y
timeout 300 python tools/dbg/srconv_dbg.py > gpurun_out/r02y/sr.txt 2>&1; cat gpurun_out/r02y/sr.txt
