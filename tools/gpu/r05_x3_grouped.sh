#!/bin/bash
# grouped weight gradients of the split-bf16 mode: tests + the mode's bench line + per-kernel totals
out=gpurun_out/${1:-r05x3g}; mkdir -p $out
timeout 900 python -m pytest tests/test_gemm.py -x -q -m gpu -k "not forced_tile" > $out/test_gemm.txt 2>&1; tail -3 $out/test_gemm.txt
timeout 900 python bench.py --dtype f32x3 --no-cpu-baseline --no-parity-mode > $out/bench_x3.json 2> $out/bench_x3.err; cut -c1-200 $out/bench_x3.json
timeout 1200 python -m pytest tests/test_dacs.py -x -q -m gpu -k "x3" > $out/test_dacs.txt 2>&1; tail -3 $out/test_dacs.txt
BENCH_ARGS="--dtype f32x3" bash tools/gpu/r05_stats.sh ${1:-r05x3g}_stats | head -16
