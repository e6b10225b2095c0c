#!/bin/bash
mkdir -p gpurun_out/sanity
timeout 600 python tools/train_sanity.py 80 > gpurun_out/sanity/out.txt 2> gpurun_out/sanity/err; tail -2 gpurun_out/sanity/err; cat gpurun_out/sanity/out.txt
