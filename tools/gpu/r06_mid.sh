#!/bin/bash
# round 6, mid-round measurements: parity record (accuracy block of the bench line + the bf16 reference-fixture figures), GEMM traffic by
# kernel instance (two PMC passes + the bench's launch record), graph-timed HBM kernels, the new RCCL single-rank tests
out=gpurun_out/${1:-r06mid}; mkdir -p $out
timeout 600 python -m pytest tests/test_parallel.py -x -q -m gpu -p no:cacheprovider > $out/tests_parallel.txt 2>&1; tail -3 $out/tests_parallel.txt
bash tools/gpu/parity.sh ${1:-r06mid}/par > $out/parity_stdout.txt 2>&1; tail -22 $out/parity_stdout.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-parity-mode"
CMDA_BENCH_GEMM_LOG=$PWD/$out/gemm_log.json rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 $B --no-graph > $out/pmc_fetch.json 2> $out/err3
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 bench.py --steps 1 --warmup 1 $B --no-graph > $out/pmc_write.json 2> $out/err4
python tools/pmc_gemm_instances.py $out/pmc_fetch $out/pmc_write $out/gemm_log.json $out/gemm_traffic_by_instance.txt
rm -rf $out/pmc_fetch $out/pmc_write
python tools/hbm_bench.py --batch 4 > $out/hbm_bench_b4.txt 2> $out/err_hbm4; tail -50 $out/hbm_bench_b4.txt
python tools/hbm_bench.py --batch 8 > $out/hbm_bench_b8.txt 2> $out/err_hbm8
