#!/bin/bash
out=gpurun_out/${1:-r03d}
mkdir -p $out
timeout 900 python -m pytest tests/test_gemm.py tests/test_dacs.py -x -q -m gpu > $out/tests.log 2>&1; tail -3 $out/tests.log
python bench.py --no-cpu-baseline --no-parity-mode > $out/bench.json 2> $out/err1; cut -c1-200 $out/bench.json
python -c "
import json
d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['ms_per_step'], r['launches_per_step'], r['gemm_ms_per_step'], r['achieved'])
"
bash tools/gpu/r03_trace.sh ${1:-r03d}
timeout 600 python tools/lanes_timeline.py > $out/lanes_timeline.txt 2> $out/err4; tail -14 $out/lanes_timeline.txt
