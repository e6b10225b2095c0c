#!/bin/bash
out=gpurun_out/${1:-r03g}
mkdir -p $out
timeout 900 python -m pytest tests/test_gemm.py -x -q -m gpu > $out/tests_gemm.log 2>&1; tail -2 $out/tests_gemm.log
python tools/gemm_bench.py --big > $out/gemm_big.txt 2>&1; cat $out/gemm_big.txt
python bench.py --no-cpu-baseline --no-parity-mode > $out/bench.json 2> $out/err1; cut -c1-200 $out/bench.json
