#!/bin/bash
# A/B on one box: grouped weight-gradient launches on eight waves; pair launches off; then the GPU suite once
out=gpurun_out/${1:-r04v}; mkdir -p $out
b() { timeout 600 python bench.py --no-cpu-baseline --no-parity-mode > $out/bench_$1.json 2> $out/err_$1; echo "$1 $(grep -o '"ms_per_step": [0-9.]*' $out/bench_$1.json)"; }
b base
CMDA_GROUPED_NW8=1 b g8
CMDA_GEMM_PAIR=0 b nopair
CMDA_GROUPED_NW8=1 CMDA_GEMM_PAIR=0 b g8_nopair
b base2
CMDA_TEST_MARGINS=$out/margins.jsonl timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $out/tests.log 2>&1; tail -4 $out/tests.log
