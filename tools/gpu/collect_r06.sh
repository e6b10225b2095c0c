#!/bin/bash
# The round's judged artefacts for the default bench line (configs[3], 2+2 samples), collected ONCE on the final commit: the bench line
# itself (with the two fp32-storage child runs and the cpu baseline), rocprofv3 kernel statistics of the same command (graph replay and
# eager launches), the PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy: one pass each) folded per family and per GEMM instance, lane
# timeline, parity record (accuracy block), large-GEMM / small-GEMM / HBM-kernel micro-benchmarks, the split-bf16 mode's eager
# statistics, the single-rank reducer line (virtual two-way RS + AG on RCCL), the supervised line.
# usage: gpurun -- 'bash tools/gpu/collect_r06.sh <tag>'; then tools/publish_profiles.sh <tag> r06 copies the summaries to profiles/
tag=${1:-r06final}
out=gpurun_out/$tag
mkdir -p $out
python bench.py > $out/bench.json 2> $out/err_bench
cut -c1-200 $out/bench.json
alg=$(python -c "import json;print(json.loads(open('$out/bench.json').read().strip().splitlines()[-1])['roofline']['algorithmic_mb_per_launch'])")
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-parity-mode"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_graph -- python3 bench.py --steps 3 --warmup 1 $B > $out/bench_prof_graph.json 2> $out/err1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_eager -- python3 bench.py --steps 3 --warmup 1 $B --no-graph > $out/bench_prof_eager.json 2> $out/err2
CMDA_BENCH_GEMM_LOG=$PWD/$out/gemm_log.json rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 $B --no-graph > $out/pmc_fetch.json 2> $out/err3
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 bench.py --steps 1 --warmup 1 $B --no-graph > $out/pmc_write.json 2> $out/err4
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $out/pmc_busy -- python3 bench.py --steps 1 --warmup 1 $B --no-graph > $out/pmc_busy.json 2> $out/err4b
python tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write dacs $out/pmc_traffic_dacs $alg
python tools/pmc_gemm_instances.py $out/pmc_fetch $out/pmc_write $out/gemm_log.json $out/gemm_traffic_by_instance.txt | head -12
python tools/pmc_busy.py $out/pmc_busy $out/mfma_busy.txt 0 | tail -8
t=$(find $out/stats_eager -name '*kernel_trace.csv' | head -1); gzip -c $t > $out/trace_eager.csv.gz
rm -rf $out/pmc_fetch $out/pmc_write $out/pmc_busy $out/stats_*/*/*kernel_trace.csv
timeout 600 python tools/lanes_timeline.py > $out/lanes_timeline.txt 2> $out/err5
bash tools/gpu/parity.sh $tag/par > $out/parity_stdout.txt 2>&1; tail -12 $out/parity_stdout.txt
python tools/gemm_bench.py --big > $out/gemm_big.txt 2>&1
python tools/hbm_bench.py --batch 4 > $out/hbm_bench_b4.txt 2>&1
python tools/hbm_bench.py --batch 8 > $out/hbm_bench_b8.txt 2>&1
python tools/dbg/rp_bench.py > $out/small_gemm.txt 2>&1
# split-bf16 mode: eager statistics of `bench.py --dtype f32x3`
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_x3 -- python3 bench.py --dtype f32x3 --steps 2 --warmup 1 $B --no-graph > $out/bench_prof_x3.json 2> $out/err7
cp "$(ls -t $out/stats_x3/*/*kernel_stats.csv | head -1)" $out/x3_eager_kernel_stats.csv; rm -rf $out/stats_x3
# the RCCL path with ONE rank: gradient exchange armed, every bucket padded / scattered / gathered as for two ranks (virtual_ways)
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-reducer $B > $out/bench_force_reducer.json 2> $out/err6
cut -c1-200 $out/bench_force_reducer.json
timeout 600 python bench.py --workload supervised --no-cpu-baseline > $out/supervised.json 2> $out/err_sup; cut -c1-200 $out/supervised.json
ls $out
