#!/bin/bash
# A/B of a global GEMM tile_hint on the bench (same box, back to back): 0 = heuristics, 32 = two LDS stages everywhere
mkdir -p gpurun_out/hint
for h in ${HINTS:-49152 81920 163840 49152 81920 163840}; do
  CMDA_BENCH_GEMM_HINT=$h timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/hint/bench_$h.json 2> gpurun_out/hint/err_$h
  python -c "
import json;d=json.loads(open('gpurun_out/hint/bench_$h.json').read().strip().splitlines()[-1]);print('hint $h', d['ms_per_step'], d['roofline']['gemm_ms_per_step'])"
done
