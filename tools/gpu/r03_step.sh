#!/bin/bash
# tests of the GEMM family + the step: bench line, lane timeline
out=gpurun_out/${1:-r03s}
mkdir -p $out
timeout 900 python -m pytest tests/test_gemm.py tests/test_modules.py -x -q -m gpu > $out/tests.log 2>&1; tail -2 $out/tests.log
python tools/gemm_bench.py --big 2>&1 | grep -E "^---|wgrad|wpw" | head -3
python bench.py --no-cpu-baseline --no-parity-mode > $out/bench.json 2> $out/err1; cut -c1-200 $out/bench.json
timeout 600 python tools/lanes_timeline.py 2>/dev/null | tail -9
