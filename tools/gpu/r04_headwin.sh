#!/bin/bash
# kernel trace of the replayed step; which kernels run while only one lane is busy (the decode-head segment)
out=gpurun_out/${1:-r04w}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode > $GRAFT_REPO_ROOT/$out/bench.json 2> $GRAFT_REPO_ROOT/$out/err
cd $GRAFT_REPO_ROOT
python tools/timeline.py $out/trace > $out/timeline.txt 2>&1; cat $out/timeline.txt
rm -rf $out/trace
