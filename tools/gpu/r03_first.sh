#!/bin/bash
# round 3, first call: the bench line of HEAD on this box, the eager kernel trace (kept, gzipped, for offline per-stage analysis),
# the copyBuffer census and the lane timeline
out=gpurun_out/r03a
mkdir -p $out
python bench.py > $out/bench.json 2> $out/err_bench
cut -c1-260 $out/bench.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $out/trace_eager -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-graph > $out/bench_prof_eager.json 2> $out/err2
python tools/copybuffer_census.py $out/trace_eager > $out/copybuffer_eager.txt 2>&1
cat $out/copybuffer_eager.txt
rocprofv3 --kernel-trace --output-format csv -d $out/trace_graph -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode > $out/bench_prof_graph.json 2> $out/err3
python tools/copybuffer_census.py $out/trace_graph > $out/copybuffer_graph.txt 2>&1
head -12 $out/copybuffer_graph.txt
for d in trace_eager trace_graph; do f=$(find $out/$d -name '*kernel_trace.csv' | head -1); gzip -c $f > $out/$d.csv.gz; rm -rf $out/$d; done
timeout 600 python tools/lanes_timeline.py > $out/lanes_timeline.txt 2> $out/err4; cat $out/lanes_timeline.txt
ls -la $out
