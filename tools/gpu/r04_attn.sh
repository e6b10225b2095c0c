#!/bin/bash
out=gpurun_out/${1:-r04z}; mkdir -p $out
for q in default 64 128 256; do
  echo "== queries per block: $q"
  if [ $q = default ]; then python tools/dbg/attn_graph_bench.py 2>&1 | grep -v amdgpu.ids; else CMDA_ATTN_QPB=$q python tools/dbg/attn_graph_bench.py 2>&1 | grep -v amdgpu.ids; fi
done | tee $out/attn.txt
