#!/bin/bash
out=gpurun_out/${1:-r04c}; mkdir -p $out
python tools/dbg/rp_bench.py > $out/rp_bench.txt 2>&1; cat $out/rp_bench.txt
python tools/gemm_bench.py --big > $out/gemm_big.txt 2>&1; cat $out/gemm_big.txt
timeout 600 python bench.py --no-cpu-baseline --no-parity-mode > $out/bench_base.json 2> $out/err0; cut -c1-180 $out/bench_base.json
timeout 900 python -m pytest tests/test_gemm.py tests/test_kernels.py -q -m gpu -x 2>&1 | tail -3
