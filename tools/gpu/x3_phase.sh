#!/bin/bash
# phase stamps of the lean split-bf16 GEMM (timing build) + the isolated table
CMDA_HIP_LIB=$PWD/build/libcmda_hip_x3timing.so python tools/dbg/x3_phase.py 2048 320 320 2>&1 | grep -v amdgpu
CMDA_HIP_LIB=$PWD/build/libcmda_hip_x3timing.so python tools/dbg/x3_phase.py 8192 320 1280 2>&1 | grep -v amdgpu | cut -c1-300
X3_FROM=0 python tools/dbg/x3_bench.py 2>&1 | grep -E "^  N[TN] +(2048|4096|8192|16384|65536|1024) x" | head -24
