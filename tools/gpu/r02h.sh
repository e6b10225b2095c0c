# the round's judged artefacts for the default bench line: kernel stats + PMC traffic + the bench line itself
mkdir -p gpurun_out/r02h
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02h/stats_graph -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02h/bench_prof_graph.json 2> gpurun_out/r02h/err1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02h/stats_eager -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > gpurun_out/r02h/bench_prof_eager.json 2> gpurun_out/r02h/err2
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r02h/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph > gpurun_out/r02h/pmc_fetch.json 2> gpurun_out/r02h/err3
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r02h/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph > gpurun_out/r02h/pmc_write.json 2> gpurun_out/r02h/err4
python tools/pmc_traffic.py gpurun_out/r02h/pmc_fetch gpurun_out/r02h/pmc_write dacs gpurun_out/r02h/pmc_traffic_dacs 9.74
rm -rf gpurun_out/r02h/pmc_fetch gpurun_out/r02h/pmc_write gpurun_out/r02h/stats_*/*/*kernel_trace.csv
python bench.py > gpurun_out/r02h/bench.json 2> gpurun_out/r02h/err5
cut -c1-300 gpurun_out/r02h/bench.json
