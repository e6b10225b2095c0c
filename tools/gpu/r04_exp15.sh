#!/bin/bash
out=gpurun_out/${1:-r04pad}; mkdir -p $out
b() { timeout 600 python bench.py --no-cpu-baseline --no-parity-mode > $out/bench_$1.json 2> $out/err_$1; echo "$1 $(grep -o '"ms_per_step": [0-9.]*' $out/bench_$1.json)"; }
b base
CMDA_GROUPED_PAD128=1.21 b pad
b base2
CMDA_GROUPED_PAD128=1.21 b pad2
CMDA_GROUPED_PAD128=1.21 timeout 900 python -m pytest tests/test_gemm.py -q -m gpu -x -k "grouped" 2>&1 | tail -2
