#!/bin/bash
out=gpurun_out/${1:-r04ce}; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels.py -q -m gpu -x -k "ce_upsample" 2>&1 | tail -2
b() { timeout 600 python bench.py --no-cpu-baseline --no-parity-mode > $out/bench_$1.json 2> $out/err_$1; echo "$1 $(grep -o '"ms_per_step": [0-9.]*' $out/bench_$1.json)"; }
b tiled
CMDA_CE_GATHER=1 b gather
b tiled2
CMDA_CE_GATHER=1 b gather2
python tools/dbg/ce_dbg.py 2>&1 | grep -v amdgpu.ids | tail -8
CMDA_CE_GATHER=1 python tools/dbg/ce_dbg.py 2>&1 | grep -v amdgpu.ids | tail -2
