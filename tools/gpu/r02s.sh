#!/bin/bash
mkdir -p gpurun_out/r02s
timeout 900 python -m pytest tests/test_parallel.py tests/test_dacs.py -x -q -m gpu > gpurun_out/r02s/tests.log 2>&1; tail -5 gpurun_out/r02s/tests.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --force-reducer --no-cpu-baseline > gpurun_out/r02s/bench_torchrun1.json 2> gpurun_out/r02s/err_torchrun
grep -v "^$" gpurun_out/r02s/err_torchrun | tail -5; cut -c1-250 gpurun_out/r02s/bench_torchrun1.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02s/bench.json 2> gpurun_out/r02s/err_bench; cut -c1-250 gpurun_out/r02s/bench.json
