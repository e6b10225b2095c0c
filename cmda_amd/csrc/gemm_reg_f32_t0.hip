// gemm_reg_f32_t0.hip -- register-staged GEMM kernel, f32, 128x128 tile (own translation unit; see gemm_reg.hip).
#include "gemm_kernels.h"

int cmda_gemm_reg_f32_t0_(const cmda_gemm_params_t& p, void* stream) { return launch_tile<float, 4, 4>(p, stream); }
