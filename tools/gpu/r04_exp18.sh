#!/bin/bash
out=gpurun_out/${1:-r04w8}; mkdir -p $out
b() { timeout 600 python bench.py --no-cpu-baseline --no-parity-mode > $out/bench_$1.json 2> $out/err_$1; echo "$1 $(grep -o '"ms_per_step": [0-9.]*' $out/bench_$1.json)"; }
b base
CMDA_LEAN_W8_MAX=2048 b w8max2048
CMDA_LEAN_W8_MAX=1024 b w8max1024
CMDA_LEAN_W8_MAX=512 b w8max512
CMDA_LEAN_W8_MAX=320 b w8max320
b base2
