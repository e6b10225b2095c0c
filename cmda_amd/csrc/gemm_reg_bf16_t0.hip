// gemm_reg_bf16_t0.hip -- register-staged GEMM kernel, bf16, 128x128 tile (own translation unit; see gemm_reg.hip).
#include "gemm_kernels.h"

int cmda_gemm_reg_bf16_t0_(const cmda_gemm_params_t& p, void* stream) { return launch_tile<bf16_t, 4, 4>(p, stream); }
