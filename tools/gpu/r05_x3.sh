#!/bin/bash
out=gpurun_out/${1:-r05x3}; mkdir -p $out
timeout 900 python -m pytest tests/test_gemm.py -x -q -m gpu > $out/test_gemm.txt 2>&1; tail -3 $out/test_gemm.txt
timeout 600 python tools/dbg/x3_bench.py > $out/x3_bench.txt 2>&1; grep -v amdgpu.ids $out/x3_bench.txt
