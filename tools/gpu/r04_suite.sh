#!/bin/bash
# usage: r04_suite.sh <tag>   -- the whole GPU suite once (margin log on) + the bench line (no parity child, no cpu baseline)
tag=${1:-r04s}; out=gpurun_out/$tag; mkdir -p $out
CMDA_TEST_MARGINS=$out/margins.jsonl timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $out/tests.log 2>&1; tail -4 $out/tests.log
timeout 600 python bench.py --no-cpu-baseline --no-parity-mode > $out/bench.json 2> $out/bench.err; cut -c1-200 $out/bench.json
