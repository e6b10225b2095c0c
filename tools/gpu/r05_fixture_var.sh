#!/bin/bash
# run-to-run / configuration spread of the reference-step fixture test in the split-bf16 mode (worst gradient fingerprint per iteration)
for cfg in "CMDA_X3_BIG_INTENSITY=100 CMDA_BN_FUSED_STATS=1" "CMDA_X3_BIG_INTENSITY=100 CMDA_BN_FUSED_STATS=0" "CMDA_X3_BIG_INTENSITY=0 CMDA_BN_FUSED_STATS=1" "CMDA_X3_BIG_INTENSITY=0 CMDA_BN_FUSED_STATS=0"; do
  for r in 1 2 3; do
    echo "== $cfg run $r"
    env $cfg timeout 600 python -m pytest tests/test_dacs.py -q -m gpu -s -k "reference_fixture_gpu and x3" 2>&1 | grep -E "^iteration|passed|failed" | sed 's/losses.*pseudo/pseudo/'
  done
done
