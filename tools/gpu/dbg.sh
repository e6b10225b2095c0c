CMDA_HIP_LIB=build/libcmda_hip_timing.so python tools/gemm_phase.py 2>&1 | grep -v amdgpu.ids | head -14
