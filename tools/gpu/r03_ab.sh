#!/bin/bash
# A/B of the deferred / grouped weight gradients + the GPU tests that cover them
out=gpurun_out/r03b
mkdir -p $out
timeout 900 python -m pytest tests/test_gemm.py tests/test_dacs.py tests/test_parallel.py -x -q -m gpu > $out/tests.log 2>&1; tail -3 $out/tests.log
CMDA_GEMM_DEFER=0 python bench.py --no-cpu-baseline > $out/bench_nodefer.json 2> $out/err0; cut -c1-200 $out/bench_nodefer.json
python bench.py --no-cpu-baseline > $out/bench_defer.json 2> $out/err1; cut -c1-200 $out/bench_defer.json
python -c "
import json
for f in ('nodefer','defer'):
    d=json.loads(open('$out/bench_%s.json'%f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f, d['ms_per_step'], r['launches_per_step'], r['gemm_ms_per_step'], r['achieved'])
"
timeout 600 python tools/lanes_timeline.py > $out/lanes_timeline.txt 2> $out/err4; tail -16 $out/lanes_timeline.txt
