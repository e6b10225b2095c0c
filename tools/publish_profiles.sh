#!/bin/bash
# copy the summaries of one tools/gpu/collect_r06.sh (collect_r05.sh) run from gpurun_out/<tag>/ into profiles/ (tracked)
tag=${1:-r05final}
R=${2:-r05}
src=gpurun_out/$tag
cp $src/bench.json profiles/${R}_dacs_bench.json
cp "$(ls -t $src/stats_graph/*/*kernel_stats.csv | head -1)" profiles/${R}_dacs_graph_kernel_stats.csv   # (newest: a re-run into the same tag leaves the older PID's files)
cp "$(ls -t $src/stats_eager/*/*kernel_stats.csv | head -1)" profiles/${R}_dacs_eager_kernel_stats.csv
cp $src/bench_prof_graph.json profiles/${R}_dacs_graph_profiled.json
cp $src/bench_prof_eager.json profiles/${R}_dacs_eager_profiled.json
cp $src/pmc_traffic_dacs.json profiles/pmc_traffic_dacs.json
cp $src/pmc_traffic_dacs.txt profiles/${R}_dacs_pmc_traffic.txt
cp $src/lanes_timeline.txt profiles/${R}_lanes_timeline.txt
grep -v amdgpu.ids $src/gemm_big.txt > profiles/${R}_gemm_big.txt
if [ -f $src/hbm_bench_b4.txt ]; then { echo "# python tools/hbm_bench.py --batch 4 / 8: GRAPH-TIMED launches (each figure includes the ~1.8-us dependent-launch boundary), algorithmic bytes"; echo "## --batch 4"; grep -v amdgpu.ids $src/hbm_bench_b4.txt; echo "## --batch 8"; grep -v amdgpu.ids $src/hbm_bench_b8.txt; } > profiles/${R}_hbm_bench.txt; else grep -v amdgpu.ids $src/hbm_bench.txt > profiles/${R}_hbm_bench.txt; fi
cp $src/gemm_traffic_by_instance.txt profiles/${R}_gemm_traffic_by_instance.txt 2>/dev/null
if [ -f $src/par/parity.json ]; then cp $src/par/parity.json profiles/${R}_parity.json; { grep -E "^\[|iteration" $src/par/tests_dacs.log | cut -c1-600; grep -E "^\[" $src/par/tests_fullsize.log | cut -c1-400; grep -v amdgpu $src/par/bf16_error_split.txt | tail -8; } > profiles/${R}_parity.txt; grep "bf16 vs reference" $src/par/tests_dacs.log > profiles/${R}_reference_step_fixture_bf16.txt; fi
grep -v amdgpu.ids $src/small_gemm.txt > profiles/${R}_small_gemm.txt 2>/dev/null
cp $src/bench_force_reducer.json profiles/${R}_dacs_force_reducer.json 2>/dev/null
cp $src/supervised.json profiles/${R}_supervised_bench.json 2>/dev/null
cp $src/mfma_busy.txt profiles/${R}_dacs_mfma_busy.txt 2>/dev/null
cp $src/x3_eager_kernel_stats.csv profiles/${R}_x3_eager_kernel_stats.csv 2>/dev/null
grep -v amdgpu.ids $src/x3_gemm.txt > profiles/${R}_x3_gemm.txt 2>/dev/null
ls -la profiles | tail -22
