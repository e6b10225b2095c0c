#!/bin/bash
# run-to-run / configuration spread of the reference-step fixture test (worst gradient fingerprint per iteration); MODE = f32 | x3
MODE=${MODE:-x3}
for cfg in ${CFGS:-"CMDA_BN_FUSED_STATS=1" "CMDA_BN_FUSED_STATS=0"}; do
  for r in 1 2 3; do
    echo "== $MODE $cfg run $r"
    env $cfg timeout 600 python -m pytest tests/test_dacs.py -q -m gpu -s -k "reference_fixture_gpu and $MODE" 2>&1 | grep -E "^iteration|passed|failed" | sed 's/losses.*pseudo/pseudo/'
  done
done
