#!/bin/bash
# eager kernel statistics + lanes timeline of the current build
tag=${1:-r04p}; out=gpurun_out/$tag; mkdir -p $out
timeout 600 python bench.py --no-cpu-baseline --no-parity-mode > $out/bench.json 2> $out/bench.err; cut -c1-200 $out/bench.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_eager -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode --no-graph > $out/bench_prof_eager.json 2> $out/err2
f=$(find $out/stats_eager -name '*kernel_stats.csv' | head -1); cp $f $out/eager_kernel_stats.csv
t=$(find $out/stats_eager -name '*kernel_trace.csv' | head -1); gzip -c $t > $out/trace_eager.csv.gz; rm -rf $out/stats_eager
timeout 600 python tools/lanes_timeline.py > $out/lanes_timeline.txt 2> $out/err5; cat $out/lanes_timeline.txt
python tools/gemm_bench.py --big 2>&1 | head -14 > $out/gemm_big.txt; cat $out/gemm_big.txt
