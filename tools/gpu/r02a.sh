mkdir -p gpurun_out/r02a
./build/mfma_peak > gpurun_out/r02a/mfma_peak.txt 2>&1
./build/store_bw > gpurun_out/r02a/store_bw.txt 2>&1
python bench.py --steps 5 --warmup 2 > gpurun_out/r02a/bench_dacs.json 2> gpurun_out/r02a/bench_dacs.err
tail -3 gpurun_out/r02a/bench_dacs.err
cat gpurun_out/r02a/bench_dacs.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02a/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02a/bench_prof.json 2> gpurun_out/r02a/bench_prof.err
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
