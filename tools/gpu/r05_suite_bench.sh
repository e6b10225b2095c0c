#!/bin/bash
# round 5: the whole GPU suite with the margin log, smoke, the default bench line and the split-bf16 line
out=gpurun_out/${1:-r05suite}; mkdir -p $out; rm -f $out/margins.jsonl
CMDA_TEST_MARGINS=$PWD/$out/margins.jsonl timeout 1500 python -m pytest tests/ -x -q -m gpu > $out/tests.log 2>&1; tail -5 $out/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -2 $out/smoke.txt
timeout 900 python bench.py --no-parity-mode > $out/bench.json 2> $out/bench.err; tail -c 3000 $out/bench.json
timeout 600 python bench.py --dtype f32x3 --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode > $out/bench_x3.json 2> $out/bench_x3.err; python - $out <<'PY'
import json, sys
for n in ('bench_x3.json',):
    try:
        d = json.loads([l for l in open(sys.argv[1] + '/' + n) if l.startswith('{')][-1])
        print(n, d['ms_per_step'], 'ms/step', d['value'], d['unit'])
    except Exception as e:
        print(n, 'failed', e)
PY
