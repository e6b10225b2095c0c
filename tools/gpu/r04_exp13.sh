#!/bin/bash
out=gpurun_out/${1:-r04ln2}; mkdir -p $out
timeout 900 python -m pytest tests/test_gemm.py -q -m gpu -x -k "ln_gemm" 2>&1 | tail -2
python tools/dbg/enc_scaling.py 2>&1 | grep -v amdgpu.ids | grep "save=True"
CMDA_LN_GEMM=0 python tools/dbg/enc_scaling.py 2>&1 | grep -v amdgpu.ids | grep "save=True"
b() { timeout 600 python bench.py --no-cpu-baseline --no-parity-mode > $out/bench_$1.json 2> $out/err_$1; echo "$1 $(grep -o '"ms_per_step": [0-9.]*' $out/bench_$1.json)"; }
b fused
CMDA_LN_GEMM=0 b two
b fused2
CMDA_LN_GEMM=0 b two2
