#!/bin/bash
# DACS parity tests (incl. graph replay) + the default bench line
out=gpurun_out/${1:-r05step}; mkdir -p $out
timeout 1500 python -m pytest tests/test_dacs.py tests/test_parallel.py -x -q -m gpu -k "not full_depth" > $out/tests.log 2>&1; tail -4 $out/tests.log
timeout 900 python bench.py --no-parity-mode --no-cpu-baseline ${BENCH_ARGS} > $out/bench.json 2> $out/bench.err; python - $out <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1] + '/bench.json') if l.startswith('{')][-1])
print('bench:', d['ms_per_step'], 'ms/step', d['value'], d['unit'], '| mit', d['roofline'].get('mit_blocks', {}).get('frac'), '| launches', d['roofline']['launches_per_step'])
PY
