#!/bin/bash
out=gpurun_out/${1:-r03r}
mkdir -p $out
timeout 900 python -m pytest tests/test_kernels.py tests/test_modules.py -x -q -m gpu > $out/tests_a.log 2>&1; tail -2 $out/tests_a.log
timeout 1500 python -m pytest tests/test_dacs.py tests/test_fullsize.py -x -q -m gpu -s > $out/tests_b.log 2>&1; grep -E "^\[|passed|failed|^E " $out/tests_b.log | tail -24
python bench.py --no-cpu-baseline --no-parity-mode > $out/bench.json 2> $out/err1; cut -c1-200 $out/bench.json
