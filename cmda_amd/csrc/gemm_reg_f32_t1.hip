// gemm_reg_f32_t1.hip -- register-staged GEMM kernel, f32, 128x64 tile (own translation unit; see gemm_reg.hip).
#include "gemm_kernels.h"

int cmda_gemm_reg_f32_t1_(const cmda_gemm_params_t& p, void* stream) { return launch_tile<float, 4, 2>(p, stream); }
