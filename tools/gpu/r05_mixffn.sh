#!/bin/bash
# round 5: fused MixFFN forward -- GPU parity of the kernel test + the graph-timed A/B against the four launches
out=gpurun_out/${1:-r05mixffn}; mkdir -p $out
timeout 600 python -m pytest tests/test_kernels.py -x -q -m gpu -k "mixffn" > $out/test.txt 2>&1; tail -5 $out/test.txt
timeout 600 python tools/dbg/mixffn_bench.py > $out/bench.txt 2>&1; grep -v amdgpu.ids $out/bench.txt
