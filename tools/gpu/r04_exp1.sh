#!/bin/bash
out=gpurun_out/${1:-r04b}; mkdir -p $out
python tools/dbg/rp_bench.py > $out/rp_bench.txt 2>&1; cat $out/rp_bench.txt
timeout 600 python bench.py --no-cpu-baseline --no-parity-mode > $out/bench_base.json 2> $out/err0; cut -c1-180 $out/bench_base.json
CMDA_LANES=enc,hw timeout 600 python bench.py --no-cpu-baseline --no-parity-mode > $out/bench_hw.json 2> $out/err1; cut -c1-180 $out/bench_hw.json; tail -3 $out/err1
