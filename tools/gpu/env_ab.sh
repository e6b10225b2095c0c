#!/bin/bash
# same-box A/B of an environment switch over the default bench line: bash tools/gpu/env_ab.sh VAR a b [runs]   (ms per step, alternating runs)
V=$1; A=$2; B=$3; N=${4:-3}
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['losses'])"; }
for i in $(seq $N); do run $V=$A; run $V=$B; done
