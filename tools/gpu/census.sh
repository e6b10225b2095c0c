#!/bin/bash
mkdir -p gpurun_out/census
timeout 600 python tools/callsites.py > gpurun_out/census/calls.txt 2> gpurun_out/census/err; tail -2 gpurun_out/census/err; cat gpurun_out/census/calls.txt
