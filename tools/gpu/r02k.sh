mkdir -p gpurun_out/r02k
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02k/stats_eager2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > gpurun_out/r02k/bench_prof_eager.json 2> gpurun_out/r02k/err2
rm -f gpurun_out/r02k/stats_*/*/*kernel_trace.csv
