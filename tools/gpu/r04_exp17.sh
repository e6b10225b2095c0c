#!/bin/bash
# the three small items (BatchNorm finalize on 32 lanes per channel, exact CE receive window, lean patch-store epilogue): unit tests + step
out=gpurun_out/${1:-r04s3}; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels.py tests/test_modules.py tests/test_gemm.py -q -m gpu -x 2>&1 | tail -2
b() { timeout 600 python bench.py --no-cpu-baseline --no-parity-mode > $out/bench_$1.json 2> $out/err_$1; echo "$1 $(grep -o '"ms_per_step": [0-9.]*' $out/bench_$1.json)"; }
b new
b new2
python tools/dbg/ce_dbg.py 2>&1 | grep -v amdgpu.ids | tail -1
