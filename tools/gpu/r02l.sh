#!/bin/bash
# fused student passes: parity on the GPU + bench A/B over the lane sets
mkdir -p gpurun_out/r02l
timeout 900 python -m pytest tests/test_dacs.py tests/test_modules.py -x -q -m gpu > gpurun_out/r02l/tests.log 2>&1
tail -3 gpurun_out/r02l/tests.log
for lanes in enc enc,T none; do
  CMDA_BENCH_LANES=$lanes timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r02l/bench_$lanes.json 2> gpurun_out/r02l/bench_$lanes.err
  tail -c 600 gpurun_out/r02l/bench_$lanes.json
done
