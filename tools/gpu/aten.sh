#!/bin/bash
mkdir -p gpurun_out/aten
timeout 600 python tools/aten_census.py > gpurun_out/aten/out.txt 2> gpurun_out/aten/err; tail -3 gpurun_out/aten/err; cat gpurun_out/aten/out.txt
