#!/bin/bash
# round-end check on one box: GPU test suite, smoke(), the supervised bench line, then the judged artefacts (collect_r03.sh)
tag=${1:-r03final}
mkdir -p gpurun_out/final
timeout 2700 python -m pytest tests -x -q -m gpu > gpurun_out/final/tests.log 2>&1; tail -3 gpurun_out/final/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; tail -1 gpurun_out/final/smoke.log
timeout 600 python bench.py --workload supervised --no-cpu-baseline --no-parity-mode > gpurun_out/final/supervised.json 2> gpurun_out/final/err_sup; cut -c1-220 gpurun_out/final/supervised.json
bash tools/gpu/collect_r03.sh $tag > gpurun_out/final/collect.log 2>&1; tail -3 gpurun_out/final/collect.log
cut -c1-220 gpurun_out/$tag/bench.json
