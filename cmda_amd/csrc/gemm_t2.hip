// gemm_t2.hip -- the 64x64 instantiations of the LDS-DMA GEMM kernel (own translation unit: the six operand-mode
// variants of a tile compile in parallel with the other tiles).  Templates: gemm_kernels.h; chosen by launch_dtype in gemm.hip.
#include "gemm_kernels.h"

int cmda_gemm_glds_t2_(const cmda_gemm_params_t& p, void* stream) { return launch_glds<2, 2>(p, stream); }

// two problems in one grid (cmda_gemm_pair): K-contiguous A, B K-contiguous (forward) or K-strided (data gradients)
int cmda_gemm_glds_pair_t2_(const cmda_gemm_params_t& p0, const cmda_gemm_params_t& p1, void* stream) {
  if (p0.b_kstrided) return launch_glds_pair<2, 2, false, true>(p0, p1, stream);
  return launch_glds_pair<2, 2, false, false>(p0, p1, stream);
}
