#!/bin/bash
mkdir -p gpurun_out/suite
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/suite/tests.log 2>&1; tail -5 gpurun_out/suite/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/suite/smoke.log 2>&1; tail -2 gpurun_out/suite/smoke.log
