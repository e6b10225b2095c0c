# concurrency lanes in the graph: correctness (graph test x3) + bench with / without lanes
mkdir -p gpurun_out/r02d
for i in 1 2 3; do python -m pytest tests/test_dacs.py -m gpu -x -q -k graph 2>&1 | tail -2; done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02d/bench_lanes.json 2> gpurun_out/r02d/bench_lanes.err
tail -3 gpurun_out/r02d/bench_lanes.err; cut -c1-400 gpurun_out/r02d/bench_lanes.json
CMDA_BENCH_NO_LANES=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02d/bench_nolanes.json 2> gpurun_out/r02d/bench_nolanes.err
tail -3 gpurun_out/r02d/bench_nolanes.err; cut -c1-400 gpurun_out/r02d/bench_nolanes.json
