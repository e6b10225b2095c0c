#!/bin/bash
out=gpurun_out/${1:-r03p}
mkdir -p $out
nproc; free -g | head -2
timeout 1500 python -m pytest tests/test_dacs.py -x -q -m gpu -s -k "full_depth_512 or fresh_masks" > $out/tests_dacs_new.log 2>&1; grep -E "^\[|iteration|passed|failed|Error|assert" $out/tests_dacs_new.log | tail -30
timeout 900 python -m pytest tests/test_fullsize.py -x -q -m gpu -s -k "mit_b5_daformer_512" > $out/tests_fullsize.log 2>&1; grep -E "^\[|passed|failed|Error|assert" $out/tests_fullsize.log | tail -12
