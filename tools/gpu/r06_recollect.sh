#!/bin/bash
# round 6: the artefacts that depend on the SCHEDULE re-collected after a scheduling change (kernels unchanged: the PMC passes, micro-benchmarks
# and parity record of collect_r06.sh stay valid) -- bench line, graph / eager kernel statistics, lane timeline, reducer line; + the new tests
tag=${1:-r06final2}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests/test_dacs.py -x -q -m gpu -s -p no:cacheprovider -k "overlapped_optimizer" 2>&1 | grep -E "passed|failed|last-step|assert" | tail -4
python bench.py > $out/bench.json 2> $out/err_bench
cut -c1-200 $out/bench.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-parity-mode"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_graph -- python3 bench.py --steps 3 --warmup 1 $B > $out/bench_prof_graph.json 2> $out/err1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_eager -- python3 bench.py --steps 3 --warmup 1 $B --no-graph > $out/bench_prof_eager.json 2> $out/err2
rm -rf $out/stats_*/*/*kernel_trace.csv
timeout 600 python tools/lanes_timeline.py > $out/lanes_timeline.txt 2> $out/err5; tail -5 $out/lanes_timeline.txt
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-reducer $B > $out/bench_force_reducer.json 2> $out/err6
cut -c1-200 $out/bench_force_reducer.json
