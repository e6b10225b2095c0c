#!/bin/bash
# copy the summaries of one tools/gpu/collect_r03.sh run from gpurun_out/<tag>/ into profiles/ (tracked)
tag=${1:-r03final}
src=gpurun_out/$tag
cp $src/bench.json profiles/r03_dacs_bench.json
cp "$(ls -t $src/stats_graph/*/*kernel_stats.csv | head -1)" profiles/r03_dacs_graph_kernel_stats.csv   # (newest: a re-run into the same tag leaves the older PID's files)
cp "$(ls -t $src/stats_eager/*/*kernel_stats.csv | head -1)" profiles/r03_dacs_eager_kernel_stats.csv
cp $src/bench_prof_graph.json profiles/r03_dacs_graph_profiled.json
cp $src/bench_prof_eager.json profiles/r03_dacs_eager_profiled.json
cp $src/pmc_traffic_dacs.json profiles/pmc_traffic_dacs.json
cp $src/pmc_traffic_dacs.txt profiles/r03_dacs_pmc_traffic.txt
cp $src/lanes_timeline.txt profiles/r03_lanes_timeline.txt
grep -v amdgpu.ids $src/gemm_big.txt > profiles/r03_gemm_big.txt
grep -v amdgpu.ids $src/hbm_bench.txt > profiles/r03_hbm_bench.txt
ls -la profiles | tail -14
