#!/bin/bash
# eager kernel trace of the current build (kept gzipped for offline analysis)
out=gpurun_out/${1:-r03c}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $out/trace_eager -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-graph > $out/bench_prof_eager.json 2> $out/err2
f=$(find $out/trace_eager -name '*kernel_trace.csv' | head -1); gzip -c $f > $out/trace_eager.csv.gz; rm -rf $out/trace_eager
ls -la $out
