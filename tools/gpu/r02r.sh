#!/bin/bash
mkdir -p gpurun_out/r02r
for b in 8 4 2; do timeout 300 python tools/hbm_bench.py --batch $b > gpurun_out/r02r/hbm_b$b.txt 2> gpurun_out/r02r/err_$b; done
cat gpurun_out/r02r/hbm_b8.txt
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --force-reducer --no-cpu-baseline > gpurun_out/r02r/bench_torchrun1.json 2> gpurun_out/r02r/err_torchrun
tail -3 gpurun_out/r02r/err_torchrun; cut -c1-250 gpurun_out/r02r/bench_torchrun1.json
CMDA_BENCH_LANES=enc timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02r/bench.json 2> gpurun_out/r02r/err_bench; cut -c1-250 gpurun_out/r02r/bench.json
