#!/bin/bash
mkdir -p gpurun_out/r02u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02u/trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02u/bench.json 2> gpurun_out/r02u/err
python tools/timeline.py gpurun_out/r02u/trace > gpurun_out/r02u/timeline.txt 2>&1
cat gpurun_out/r02u/timeline.txt
ls -la gpurun_out/r02u/trace/*/
gzip gpurun_out/r02u/trace/*/*kernel_trace.csv
