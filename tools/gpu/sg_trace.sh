#!/bin/bash
out=gpurun_out/sg
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $out/tr -- python3 tools/dbg/small_gemm_trace.py 0 -1 > $out/log 2>&1
python tools/trace_stats.py $out/tr 10 --runs | grep gemm
rm -rf $out/tr
