#!/bin/bash
out=gpurun_out/${1:-r04j}; mkdir -p $out
timeout 600 python bench.py --no-cpu-baseline --no-parity-mode > $out/bench.json 2> $out/err0; cut -c1-180 $out/bench.json
timeout 600 python tools/lanes_timeline.py > $out/lanes_timeline.txt 2> $out/err5; cat $out/lanes_timeline.txt
