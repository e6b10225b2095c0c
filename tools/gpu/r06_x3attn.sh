#!/bin/bash
# round 6: split-bf16 fused attention -- kernel parity on the GPU, the x3 DACS tests, same-box A/B of the x3 bench line; + the bf16 line with
# the postponed overlapped update
out=gpurun_out/${1:-r06x3}; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels.py -x -q -m gpu -p no:cacheprovider -k "fused_attention" > $out/tests_attn.txt 2>&1; grep -E "passed|failed|assert" $out/tests_attn.txt | tail -3
timeout 1500 python -m pytest tests/test_dacs.py -x -q -m gpu -s -p no:cacheprovider -k "x3 or overlapped" > $out/tests_dacs_x3.txt 2>&1; grep -E "^\[x3|passed|failed|assert|last-step" $out/tests_dacs_x3.txt | cut -c1-300 | tail -8
runx() { echo -n "$1: "; env $1 python bench.py --dtype f32x3 --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['losses'])"; }
{ for i in 1 2; do runx CMDA_ATTN_X3=0; runx CMDA_ATTN_X3=1; done; } 2>&1 | tee $out/x3_ab.txt
bash tools/gpu/env_ab.sh CMDA_OPT_OVERLAP 0 1 3 | tee $out/overlap_ab.txt
