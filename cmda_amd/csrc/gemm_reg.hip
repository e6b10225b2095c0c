// gemm_reg.hip -- dispatch of the register-staged GEMM kernel: the exact-fp32 parity mode and the bf16 operand views the
// LDS-DMA path does not take (transposed-conv zero insertion, reflection padding, unaligned chunks).  The instantiations
// live in gemm_reg_{f32,bf16}_t{0,1,2}.hip, one (dtype, tile) per translation unit.  Templates: gemm_kernels.h.
#include "gemm_kernels.h"

int cmda_gemm_reg_(const cmda_gemm_params_t& p, int tile, void* stream) {
  const int t = (tile == 3) ? 0 : (tile == 4 ? 2 : tile);  // the 256x256 tile exists on the LDS-DMA path only
  if (p.dtype == CMDA_F32X3) {
    // the encoders' Linear layers / data gradients (small grids: the 64 x 64 and 128 x 64 tile choices): the LDS-DMA instance
    // ... and every weight gradient it takes (split-K over the tokens chosen by the kernel's launcher, bias gradient fused)
    if ((tile == 1 || tile == 2 || tile == 4 || p.a_kstrided || (p.tile_hint > 0 && (p.tile_hint & 65536))) && !p.colstats && cmda_gemm_x3_lean_ok_(p))
      return cmda_gemm_x3_lean_(p, stream);   // (tile_hint bit 16: the lean kernel whatever the tile choice -- tuning A/B)
    return cmda_gemm_x3_(p, tile, stream);
  }
  if (p.dtype == CMDA_F32) return t == 0 ? cmda_gemm_reg_f32_t0_(p, stream) : t == 1 ? cmda_gemm_reg_f32_t1_(p, stream) : cmda_gemm_reg_f32_t2_(p, stream);
  return t == 0 ? cmda_gemm_reg_bf16_t0_(p, stream) : t == 1 ? cmda_gemm_reg_bf16_t1_(p, stream) : cmda_gemm_reg_bf16_t2_(p, stream);
}
