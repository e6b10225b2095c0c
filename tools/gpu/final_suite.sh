#!/bin/bash
# usage: final_suite.sh <n> [round]  -- the driver's command on a fresh box, margin log on; the log is copied to profiles/<round>_gputest_run<n>.log,
# then the smoke entry
n=${1:-1}; R=${2:-r06}; out=gpurun_out/${R}f$n; mkdir -p $out
( echo "# python -m pytest tests -x -q -m gpu   (fresh gpurun lease, $(git rev-parse --short HEAD 2>/dev/null || echo snapshot), $(date -u +%FT%TZ))"
  CMDA_TEST_MARGINS=$out/margins.jsonl timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 ) > $out/tests.log
tail -3 $out/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
