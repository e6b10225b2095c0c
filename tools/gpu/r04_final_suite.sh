#!/bin/bash
# usage: r04_final_suite.sh <n>  -- the driver's command on a fresh box, margin log on; the log is copied to profiles/r04_gputest_run<n>.log
n=${1:-1}; out=gpurun_out/r04f$n; mkdir -p $out
( echo "# python -m pytest tests -x -q -m gpu   (fresh gpurun lease, $(git rev-parse --short HEAD 2>/dev/null || echo snapshot), $(date -u +%FT%TZ))"
  CMDA_TEST_MARGINS=$out/margins.jsonl timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 ) > $out/tests.log
tail -3 $out/tests.log
