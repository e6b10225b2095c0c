out=gpurun_out/r05x3busy; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $out/pmc_busy -- python3 bench.py --dtype f32x3 --steps 1 --warmup 1 --no-cpu-baseline --no-parity-mode --no-graph > $out/pmc_busy.json 2> $out/err
python tools/pmc_busy.py $out/pmc_busy $out/x3_mfma_busy.txt 0 | tail -5
rm -rf $out/pmc_busy
