#!/bin/bash
# round 6: lane sets of the EARLY-STUDENT schedule -- graph-replay parity, same-box A/B of the bench line (alternating runs), timeline
out=gpurun_out/${1:-r06lanes}; mkdir -p $out
timeout 900 python -m pytest tests/test_dacs.py -x -q -m gpu -p no:cacheprovider -k "graph_replay_matches_oracle or fresh_masks" > $out/tests_graph.txt 2>&1; grep -E "passed|failed" $out/tests_graph.txt | tail -2
run() { echo -n "$2 lanes=$1: "; env $2 CMDA_BENCH_LANES=$1 python bench.py --no-cpu-baseline --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['losses'])"; }
{ for i in 1 2 3; do run enc X=0; run enc,T,Tenc X=0; run enc,T,Tenc,wq X=0; done; } 2>&1 | tee $out/lanes_ab.txt
CMDA_LANES=enc,T,Tenc,wq timeout 600 python tools/lanes_timeline.py > $out/lanes_timeline_wq.txt 2> $out/err5; tail -44 $out/lanes_timeline_wq.txt
