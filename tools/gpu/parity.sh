#!/bin/bash
# parity figures at the bench's configuration: the three storage / GEMM modes of the full-depth 512 x 512 DACS iteration, the
# single-modality full-size model, the bf16 error split, and the graph-replay mask test
out=gpurun_out/${1:-par}; mkdir -p $out
nproc; free -g | head -2
rm -f $out/parity.json; export CMDA_PARITY_JSON=$PWD/$out/parity.json
timeout 1500 python -m pytest tests/test_dacs.py -x -q -m gpu -s -k "full_depth_512 or fresh_masks or bf16_against_reference" > $out/tests_dacs.log 2>&1; grep -E "^\[|iteration|passed|failed|Error|assert" $out/tests_dacs.log | tail -30
timeout 900 python -m pytest tests/test_fullsize.py -x -q -m gpu -s -k "mit_b5_daformer_512 or fusion_student_512" > $out/tests_fullsize.log 2>&1; grep -E "^\[|passed|failed|Error|assert" $out/tests_fullsize.log | tail -12
timeout 900 python tools/dbg/bf16_error_split.py > $out/bf16_error_split.txt 2>&1; grep -v amdgpu $out/bf16_error_split.txt | tail -12
