#!/bin/bash
# kernel statistics of the fused-pass build (eager launches: per-kernel durations without graph replay effects)
mkdir -p gpurun_out/r02m
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02m/stats_eager -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > gpurun_out/r02m/bench_prof_eager.json 2> gpurun_out/r02m/err
rm -f gpurun_out/r02m/stats_*/*/*kernel_trace.csv
