// gemm_t3.hip -- the 256x256 / 8-wave instantiations of the LDS-DMA GEMM kernel (own translation unit: compiles in
// parallel with gemm.hip).  Templates: gemm_kernels.h; chosen by launch_dtype in gemm.hip.
#include "gemm_kernels.h"

int cmda_gemm_glds_t3_(const cmda_gemm_params_t& p, void* stream) { return launch_glds<4, 8, 8>(p, stream); }
