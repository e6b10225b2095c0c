#!/bin/bash
# A/B on one box: spatial-reduction convs on the lean kernel (patch view); pair launches; split-K threshold of the sr conv
out=gpurun_out/${1:-r04y}; mkdir -p $out
b() { timeout 600 python bench.py --no-cpu-baseline --no-parity-mode > $out/bench_$1.json 2> $out/err_$1; echo "$1 $(grep -o '"ms_per_step": [0-9.]*' $out/bench_$1.json)"; }
b base
CMDA_GEMM_PAIR=0 b nopair
CMDA_SR_SPLITK_TILES=32 b sk32
CMDA_SR_SPLITK_TILES=32 CMDA_GEMM_PAIR=0 b sk32_nopair
CMDA_SR_SPLITK_TILES=16 CMDA_GEMM_PAIR=0 b sk16_nopair
CMDA_SR_SPLITK_TILES=0 CMDA_GEMM_PAIR=0 b sk0_nopair
b base2
timeout 900 python -m pytest tests/test_kernels.py tests/test_modules.py tests/test_gemm.py -q -m gpu -x 2>&1 | tail -3
