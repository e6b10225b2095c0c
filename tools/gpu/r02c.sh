# joint passes + graph: tests, eager vs graph bench, rocprof of the graph bench
mkdir -p gpurun_out/r02c
python -m pytest tests/test_dacs.py tests/test_kernels.py tests/test_modules.py tests/test_fullsize.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -25
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph > gpurun_out/r02c/bench_eager.json 2> gpurun_out/r02c/bench_eager.err
tail -3 gpurun_out/r02c/bench_eager.err; cat gpurun_out/r02c/bench_eager.json
python bench.py --steps 10 --warmup 3 > gpurun_out/r02c/bench_graph.json 2> gpurun_out/r02c/bench_graph.err
tail -3 gpurun_out/r02c/bench_graph.err; cat gpurun_out/r02c/bench_graph.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02c/prof -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02c/bench_prof.json 2> gpurun_out/r02c/bench_prof.err
tail -2 gpurun_out/r02c/bench_prof.err; cat gpurun_out/r02c/bench_prof.json
