#!/bin/bash
# sweep of the three-launch threshold of the split-bf16 mode (FLOP per byte moved by the split / accumulate passes) + the mode's DACS tests
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-parity-mode --dtype f32x3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for v in ${SWEEP:-100 0 60 150 250}; do run CMDA_X3_BIG_INTENSITY=$v; done
timeout 1200 python -m pytest tests/test_dacs.py -x -q -m gpu -k "x3" 2>&1 | tail -40
