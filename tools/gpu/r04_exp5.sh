#!/bin/bash
out=gpurun_out/${1:-r04i}; mkdir -p $out
RP_SHORT=1 python tools/dbg/rp_bench.py > $out/floor.txt 2>&1; cat $out/floor.txt
CMDA_GEMM_PAIR=0 timeout 600 python bench.py --no-cpu-baseline --no-parity-mode > $out/bench_nopair.json 2> $out/err0; cut -c1-180 $out/bench_nopair.json
timeout 600 python bench.py --no-cpu-baseline --no-parity-mode > $out/bench_pair.json 2> $out/err1; cut -c1-180 $out/bench_pair.json
timeout 900 python -m pytest tests/test_gemm.py tests/test_kernels.py tests/test_modules.py -q -m gpu -x 2>&1 | tail -3
