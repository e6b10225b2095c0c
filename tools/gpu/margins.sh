#!/bin/bash
# usage: margins.sh <tag> [runs]   -- the whole GPU suite `runs` times with the margin log on, then a bench line
tag=${1:-r05a}; runs=${2:-2}
out=gpurun_out/$tag; mkdir -p $out
for i in $(seq 1 $runs); do
  CMDA_TEST_MARGINS=$out/margins_$i.jsonl timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=15 > $out/tests_$i.log 2>&1
  tail -4 $out/tests_$i.log
done
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; cut -c1-400 $out/bench.json
