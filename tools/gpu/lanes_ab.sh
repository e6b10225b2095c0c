#!/bin/bash
# A/B of the concurrency-lane sets on the final kernels (bench line, 10 timed steps each)
run() { echo -n "lanes=$1: "; CMDA_BENCH_LANES=$1 python bench.py --no-cpu-baseline --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
run enc
run enc,T
run enc,hw
run enc,T,hw
run enc
