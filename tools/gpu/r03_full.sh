#!/bin/bash
# GPU tests (all but the slow full-size ones unless $2 = all) + bench line + lane timeline
out=gpurun_out/${1:-r03t}
mkdir -p $out
if [ "$2" = "all" ]; then
  timeout 2400 python -m pytest tests -x -q -m gpu > $out/tests.log 2>&1
else
  timeout 1500 python -m pytest tests/test_gemm.py tests/test_kernels.py tests/test_modules.py tests/test_dacs.py -x -q -m gpu -k "not full_depth" > $out/tests.log 2>&1
fi
tail -3 $out/tests.log
python bench.py --no-cpu-baseline --no-parity-mode > $out/bench.json 2> $out/err1; cut -c1-200 $out/bench.json
timeout 600 python tools/lanes_timeline.py 2>/dev/null | tail -9
