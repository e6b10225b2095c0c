#!/bin/bash
out=gpurun_out/${1:-r04tb}; mkdir -p $out
b() { timeout 600 python bench.py --no-cpu-baseline --no-parity-mode > $out/bench_$1.json 2> $out/err_$1; echo "$1 $(grep -o '"ms_per_step": [0-9.]*' $out/bench_$1.json)"; }
b base
CMDA_TILE_BIAS2=1.3 b b2_13
CMDA_TILE_BIAS2=1.7 b b2_17
CMDA_TILE_BIAS2=1.7 CMDA_TILE_BIAS1=1.4 b b2_17_b1_14
CMDA_TILE_BIAS2=0.7 b b2_07
b base2
