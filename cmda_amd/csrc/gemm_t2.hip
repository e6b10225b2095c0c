// gemm_t2.hip -- the 64x64 instantiations of the LDS-DMA GEMM kernel (own translation unit: the six operand-mode
// variants of a tile compile in parallel with the other tiles).  Templates: gemm_kernels.h; chosen by launch_dtype in gemm.hip.
#include "gemm_kernels.h"

int cmda_gemm_glds_t2_(const cmda_gemm_params_t& p, void* stream) { return launch_glds<2, 2>(p, stream); }
