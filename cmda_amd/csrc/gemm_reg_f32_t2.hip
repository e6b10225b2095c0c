// gemm_reg_f32_t2.hip -- register-staged GEMM kernel, f32, 64x64 tile (own translation unit; see gemm_reg.hip).
#include "gemm_kernels.h"

int cmda_gemm_reg_f32_t2_(const cmda_gemm_params_t& p, void* stream) { return launch_tile<float, 2, 2>(p, stream); }
