#!/bin/bash
# LayerNorm-prologue GEMM (cmda_ln_gemm): unit tests, then the step with and without it on one box
out=gpurun_out/${1:-r04ln}; mkdir -p $out
timeout 900 python -m pytest tests/test_gemm.py tests/test_abi.py -q -m gpu -x -k "ln_gemm or abi" 2>&1 | tail -3
b() { timeout 600 python bench.py --no-cpu-baseline --no-parity-mode > $out/bench_$1.json 2> $out/err_$1; echo "$1 $(grep -o '"ms_per_step": [0-9.]*' $out/bench_$1.json)"; }
b fused
CMDA_LN_GEMM=0 b two
b fused2
CMDA_LN_GEMM=0 b two2
python tools/dbg/enc_scaling.py 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_kernels.py tests/test_modules.py tests/test_gemm.py -q -m gpu -x 2>&1 | tail -3
