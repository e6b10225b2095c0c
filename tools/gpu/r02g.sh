mkdir -p gpurun_out/r02g
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python bench.py > gpurun_out/r02g/bench.json 2> gpurun_out/r02g/bench.err
tail -3 gpurun_out/r02g/bench.err; cat gpurun_out/r02g/bench.json
python bench.py --workload supervised --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | cut -c1-600
