#!/bin/bash
mkdir -p gpurun_out/timeline
timeout 600 python tools/lanes_timeline.py > gpurun_out/timeline/out.txt 2> gpurun_out/timeline/err; tail -2 gpurun_out/timeline/err; cat gpurun_out/timeline/out.txt
