#!/bin/bash
mkdir -p gpurun_out/r02w
timeout 300 python tools/dbg/walk_dbg.py > gpurun_out/r02w/walk.txt 2>&1; grep -c "bad elems 0," gpurun_out/r02w/walk.txt; grep -v "bad elems 0," gpurun_out/r02w/walk.txt | head
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r02w/tests.log 2>&1; tail -5 gpurun_out/r02w/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02w/bench.json 2> gpurun_out/r02w/err_bench; cut -c1-250 gpurun_out/r02w/bench.json
