#!/bin/bash
# module-level golden tests + the reference-step fixture test with the margin log (tools/test_margins.py)
out=gpurun_out/${1:-r05modules}; mkdir -p $out; rm -f $out/margins.jsonl
CMDA_TEST_MARGINS=$PWD/$out/margins.jsonl timeout 2400 python -m pytest tests/test_modules.py tests/test_dacs.py -q -m gpu -k "golden or reference_fixture" > $out/test.txt 2>&1; tail -30 $out/test.txt
