#!/bin/bash
# per-kernel totals of the last eager step with the fused BatchNorm statistics on / off
for v in 1 0; do
  export CMDA_BN_FUSED_STATS=$v
  echo "=== CMDA_BN_FUSED_STATS=$v"
  bash tools/gpu/stats.sh r05cs_stats$v | grep -E "last step|gemm_glds_kernel<4, 4|gemm_glds_kernel<4, 8|gemm_glds_kernel<4, 2|gemm_pp|bn_|zero_words|dw_dilated"
done
