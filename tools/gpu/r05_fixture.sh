#!/bin/bash
out=gpurun_out/${1:-r05fixture}; mkdir -p $out
timeout 1500 python -m pytest tests/test_dacs.py -x -q -m gpu -k "reference_fixture" -s > $out/test.txt 2>&1; tail -25 $out/test.txt
