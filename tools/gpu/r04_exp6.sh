#!/bin/bash
out=gpurun_out/${1:-r04k}; mkdir -p $out
CMDA_HEAD_TAIL=0 timeout 600 python bench.py --no-cpu-baseline --no-parity-mode > $out/bench_notail.json 2> $out/err0; cut -c1-180 $out/bench_notail.json
timeout 600 python bench.py --no-cpu-baseline --no-parity-mode > $out/bench_tail.json 2> $out/err1; cut -c1-180 $out/bench_tail.json; tail -3 $out/err1
timeout 900 python -m pytest tests/test_dacs.py tests/test_parallel.py -q -m gpu -x -k "graph or hook or full_width" 2>&1 | tail -3
timeout 600 python tools/lanes_timeline.py > $out/lanes_timeline.txt 2> $out/err5; cat $out/lanes_timeline.txt
