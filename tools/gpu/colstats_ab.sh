#!/bin/bash
# BatchNorm / InstanceNorm statistics in the producing GEMM's epilogue: kernel + module tests, same-box A/B of the bench line, per-kernel totals
out=gpurun_out/${1:-r05cs}; mkdir -p $out
timeout 900 python -m pytest tests/test_abi.py tests/test_gemm.py tests/test_kernels.py -x -q -m gpu -k "not forced_tile" > $out/test_kernels.txt 2>&1; tail -3 $out/test_kernels.txt
timeout 1200 python -m pytest tests/test_modules.py tests/test_dacs.py -x -q -m gpu -k "not full_depth" > $out/test_modules.txt 2>&1; tail -3 $out/test_modules.txt
run() { echo -n "$1 $2: "; env $1 python bench.py --no-cpu-baseline --no-parity-mode $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
run CMDA_BN_FUSED_STATS=1
run CMDA_BN_FUSED_STATS=0
run CMDA_BN_FUSED_STATS=1
run CMDA_BN_FUSED_STATS=0
run CMDA_BN_FUSED_STATS=1 "--dtype f32x3"
run CMDA_BN_FUSED_STATS=0 "--dtype f32x3"
bash tools/gpu/colstats_stats.sh
