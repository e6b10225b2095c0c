#!/bin/bash
# split-bf16 GEMM tests + micro-benchmark + the mode's bench line + the mode's DACS tests
out=gpurun_out/${1:-r05x3}; mkdir -p $out
timeout 900 python -m pytest tests/test_gemm.py -x -q -m gpu -k "not forced_tile" > $out/test_gemm.txt 2>&1; tail -3 $out/test_gemm.txt
X3_FROM=${X3_FROM:-99} timeout 900 python tools/dbg/x3_bench.py > $out/x3_bench.txt 2>&1; grep -v amdgpu $out/x3_bench.txt
timeout 900 python bench.py --dtype f32x3 --no-cpu-baseline --no-parity-mode > $out/bench_x3.json 2> $out/bench_x3.err; cut -c1-200 $out/bench_x3.json
timeout 900 python -m pytest tests/test_dacs.py -x -q -m gpu -k "x3" > $out/test_dacs.txt 2>&1; tail -3 $out/test_dacs.txt
