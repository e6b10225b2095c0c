#!/bin/bash
out=gpurun_out/${1:-r04o}; mkdir -p $out
timeout 600 python bench.py --no-cpu-baseline --no-parity-mode > $out/bench.json 2> $out/err0; cut -c1-180 $out/bench.json
timeout 900 python -m pytest tests/test_kernels.py tests/test_modules.py tests/test_gemm.py -q -m gpu -x 2>&1 | tail -3
python tools/hbm_bench.py --batch 8 2>&1 | grep -v amdgpu > $out/hbm_bench.txt; cat $out/hbm_bench.txt | head -60
