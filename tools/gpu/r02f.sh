mkdir -p gpurun_out/r02f
for i in 1 2; do python -m pytest tests/test_dacs.py -m gpu -x -q 2>&1 | tail -2; done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02f/bench_lanes.json 2> gpurun_out/r02f/bench_lanes.err
tail -3 gpurun_out/r02f/bench_lanes.err; cut -c1-330 gpurun_out/r02f/bench_lanes.json
