// gemm_g1.hip -- GROUPED launches of the LDS-DMA GEMM kernel on the 128x64 tile (weight-gradient operand modes).  Templates:
// gemm_kernels.h (gemm_glds_grouped_kernel); planning and dispatch: gemm_grouped.hip.
#include "gemm_kernels.h"

int cmda_gemm_grouped_t1_(const cmda_gemm_params_t* tab, const void* blk, int nblocks, int bconv, void* stream) {
  return launch_glds_grouped<4, 2>(tab, blk, nblocks, bconv, stream);
}
