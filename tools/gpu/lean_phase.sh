#!/bin/bash
# phase stamps of the bf16 lean GEMM (timing build) at the encoders' stage-3 shapes
export CMDA_HIP_LIB=$PWD/build/libcmda_hip_leantiming.so
for s in "2048 320 320" "4096 320 320" "4096 320 320 res" "4096 1280 320" "4096 320 1280 res" "4096 320 320 nn"; do python tools/dbg/lean_phase.py $s 2>&1 | grep -v amdgpu; done
