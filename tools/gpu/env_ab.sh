#!/bin/bash
# A/B of HIP runtime knobs on the default bench line (each run: 10 timed steps of the captured iteration)
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
run X=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=1
run DEBUG_HIP_KERNARG_COPY_OPT=0
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run X=1
