// gemm_reg_bf16_t2.hip -- register-staged GEMM kernel, bf16, 64x64 tile (own translation unit; see gemm_reg.hip).
#include "gemm_kernels.h"

int cmda_gemm_reg_bf16_t2_(const cmda_gemm_params_t& p, void* stream) { return launch_tile<bf16_t, 2, 2>(p, stream); }
