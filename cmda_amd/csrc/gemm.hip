// gemm.hip -- tile / split-K heuristics and dispatch of the GEMM family.  Kernel templates and their documentation:
// gemm_kernels.h; the instantiations live in gemm_t0..t3.hip (LDS-DMA tiles) and gemm_reg.hip (register-staged).
#include "gemm_kernels.h"

namespace {

template <typename T>
int launch_dtype(GemmParams& p, void* stream) {
  // Tile / split-K choice.  256 CUs want >= ~2 blocks each; never waste half a tile on N <= 64.
  const long zb = (long)p.batch * p.batch2;
  auto blocks = [&](int bm, int bn) { return (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * zb; };
  constexpr int BK = 8 * Num<T>::kChunk;
  const int nkt = (p.K + BK - 1) / BK;
  int tile;  // 0: 128x128, 1: 128x64, 2: 64x64, 3: 256x256 (8 waves)
  if (p.atomic && p.splits <= 0) {
    // accumulate-by-atomics GEMMs (weight gradients): few output tiles, very long contraction -> largest tile that
    // fits the output, then split K until the grid fills the chip
    // Every split adds one fp32 atomic per output element (chip-wide ~1.3 TB/s), so splits stay small and the tile
    // only grows to 128x128 when the output alone already has enough tiles.
    tile = 2;
    if (p.M > 64 && p.N > 64) {
      // (the im2col-view weight gradient spills on the 256x256 tile: 11.7 ms against 9.6 ms on 128x128 at batch 64)
      // 256x256: gemm_wg.hip (32x32x16 MFMA; both operands stream from beyond L2, so the tile size is the speed: the head's pointwise
      // weight gradients 256 x 1024 over 262144 pixels are 4 tiles x 64 splits, its 3x3 bottleneck 36 tiles x 7 splits)
      if (sizeof(T) == 2 && p.M >= 256 && p.N >= 256 && p.A.conv != 1 && (blocks(256, 256) >= 32 || (blocks(256, 256) >= 4 && nkt >= 1024)))
        tile = 3;
      else if (blocks(128, 128) >= 64 || (blocks(128, 128) >= 16 && nkt >= 1024)) tile = 0;  // very deep K: split further
      else if (blocks(128, 64) >= 48 && nkt >= 256) tile = 1;  // e.g. the 320x1280 MixFFN weight gradients at K >= 16 k rows (57 vs 72 us on
                                                               // 64x64 tiles); at the UDA step's K = 8192 the 64x64 tile is ahead (30.0 vs 32.5 us)
    }
    const long b = tile == 3 ? blocks(256, 256) : tile == 0 ? blocks(128, 128) : tile == 1 ? blocks(128, 64) : blocks(64, 64);
    // im2col weight gradients (very deep K, output of a few MB): let the wave-quantisation search below look as far as two
    // full waves of the 128x128 tile (144 tiles x 7 splits: 9.6 ms against 11.2 ms at 3 splits, batch 64)
    long s = ((p.B.conv == 1 ? 1024 : 512) + b - 1) / b;
    s = std::min<long>(s, std::max(1, nkt / 8));
    const long out_bytes = (long)p.M * p.N * 4 * zb;
    s = std::min<long>(s, std::max<long>(16, (32L << 20) / std::max<long>(out_bytes, 1)));  // atomic traffic bound
    if (tile == 3 && b * s < 256) s = std::min<long>((256 + b - 1) / b, std::max(1, nkt / 8));  // one block per CU: at least fill the chip once
    s = std::min<long>(s, 128);
    s = std::min<long>(s, std::max<long>(1, 65535 / std::max<long>(zb, 1)));
    s = std::max<long>(1, s);
    // wave quantisation: the grid runs in waves of `slots` resident blocks (256 CUs x blocks per CU for this tile's LDS),
    // and a block's time is ~ K / splits -- 36 tiles x 15 splits on the one-block-per-CU 256x256 kernel is three waves
    // (256 + 256 + 28) where 36 x 7 is one.  Take the fewest splits within 5 % of the best waves/splits ratio.
    {
      const long slots = 256L * (tile == 3 ? 1 : tile == 0 ? 2 : tile == 1 ? 3 : 4);
      double best = 1e30;
      for (long c = 1; c <= s; ++c) best = std::min(best, (double)((b * c + slots - 1) / slots) / (double)c);
      for (long c = 1; c <= s; ++c)
        if ((double)((b * c + slots - 1) / slots) / (double)c <= best * 1.05) {
          s = c;
          break;
        }
    }
    p.splits = (int)s;
  } else {
    if (p.splits <= 0) p.splits = 1;
    const long sp = p.splits;
    // measured with the lean epilogue (profiles/README.md, forced-tile sweep): the 128x128 tile wins from ~1 block per CU up
    // 256x256 (8 waves, one block per CU) from ~4 blocks per CU on deep contractions: 8192^3 1302 -> 1060 us, the head's
    // 3x3 bottleneck conv 2134 -> 1570 us; short-K or narrow problems lose to the 128x128 tile
    if (sizeof(T) == 2 && p.N >= 256 && p.K >= 256 && (blocks(256, 256) * sp >= 1024 || (blocks(256, 256) * sp >= 200 && p.K >= 2048))) tile = 3;  // (4096^3: one block per CU, 1113 vs 883 TFLOP/s on 128x128)
    // output-bound GEMMs with a plain epilogue (MixFFN fc1 of stages 1 / 2: 131072 x 256 x 64, 32768 x 512 x 128; the head's pointwise
    // convolution over the teacher's 98304 rows): the ping-pong kernel's bf16 row image stores 16-byte pieces of whole rows
    // (29.6 against 38.4 us, 18.4 against 24.3 us; at 16384 x 512 the 64-wide tiles are still ahead)
    else if (sizeof(T) == 2 && (p.N & 255) == 0 && p.K <= 1024 && blocks(256, 256) * sp >= 256 && !p.a_kstrided && !p.b_kstrided && p.A.conv != 1 &&
             p.B.conv != 1 && !p.res && !p.rowscale && p.beta == 0.f && !p.out_f32 && !p.atomic && p.c_patch_ow == 0)
      tile = 3;
    // N = 320 (64 x odd): the last 128-wide tile column would be half empty -- with enough blocks the 128x64 tile wins
    // (batch 64, M = 65536: q 42.2 -> 38.9 us, dfc1 110.8 -> 99.6 us; at batch 16 the 128x128 tile is still ahead)
    else if (p.N > 64 && (p.N & 127) > 0 && (p.N & 127) <= 64 && blocks(128, 64) * sp >= 2048) tile = 1;
    else if (p.N > 64 && blocks(128, 128) * sp >= 256) tile = 0;
    else if (blocks(128, 64) * sp >= 2048) tile = 1;
    else tile = 2;
    // Small grids (at most ~two waves of 128x128 tiles: the UDA step's encoder Linears, M = 2048...8192 rows): the kernel time is
    // (waves of resident blocks) x (time of one block), so pick the tile by that product.  Block time ~ c0 + c1 * k-tiles, fitted to
    // tools/gemm_sweep.py (profiles/r02_gemm_sweep.txt): 128x128 4 + 0.9 nkt us (2 blocks per CU), 128x64 3 + 0.75 nkt (3 per CU),
    // 64x64 2.5 + 0.6 nkt (4 per CU).  8192x1280x320: 128x128 26.1 -> 128x64 20.9 us; 8192x320x1280 stays on 64x64.
    if (tile != 3 && blocks(128, 128) * sp < 1024 && p.M > 64 && p.N > 64) {
      const double nk = (double)nkt;
      const double cost0 = (double)((blocks(128, 128) * sp + 511) / 512) * (4.0 + 0.9 * nk);
      // (round 4: the lean instance of the two small tiles -- gemm_lean.hip -- has a cheaper prologue / epilogue and a 0.3 us k-tile:
      // 16384 x 512 x 128 10.9 us on 64 x 64 tiles against 13.7 on 128 x 128, tools/dbg/pp_vs_lean.py)
      const bool lean = sizeof(T) == 2 && cmda_gemm_lean_ok_(p, 2);
      const double cost1 = (double)((blocks(128, 64) * sp + 767) / 768) * (lean ? 1.5 + 0.45 * nk : 3.0 + 0.75 * nk);
      const double cost2 = (double)((blocks(64, 64) * sp + 1023) / 1024) * (lean ? 1.2 + 0.3 * nk : 2.5 + 0.6 * nk);
      tile = cost0 <= cost1 && cost0 <= cost2 ? 0 : cost1 <= cost2 ? 1 : 2;
    }
  }
  if ((p.tile_hint & 15) >= 1 && (p.tile_hint & 15) <= 4 && p.tile_hint > 0) tile = (p.tile_hint & 15) - 1;  // caller's explicit tile choice (tuning sweeps, tests)
  if constexpr (sizeof(T) == 2) {
    const bool no_glds = p.tile_hint < 0;  // caller asks for the register-staged kernel (tuning sweeps)
    // LDS-DMA path: every operand mode with aligned 16-byte chunks (reflection padding and zero-inserted inputs included); operands
    // too large for 32-bit tile arithmetic stay on the register-staged kernel
    // (zero-inserted inputs, in_dil > 1 -- transposed convolutions, data gradients of strided ones --: K-contiguous im2col operands only)
    auto dma_ok = [&](const GemmView& v) {
      return v.vec_ok && (v.in_dil <= 1 || (v.conv == 1 && !v.reflect && &v == &p.A && !p.a_kstrided)) && v.R < (1L << 31) && v.Cc < (1L << 31) &&
             (!v.conv || (v.H < 32768 && v.W < 32768)) && (v.conv || (v.ld % 8) == 0) && (v.Cc % 8) == 0 &&
             (v.conv != 2 || ((v.KW * v.C) % 64 == 0 && v.KH == v.stride && v.KW == v.stride && v.pad == 0 && v.dil == 1 &&
                              v.H == v.OH * v.stride && v.W == v.OW * v.stride));
    };
    const bool ac = p.A.conv == 1, bc = p.B.conv == 1, aks = p.a_kstrided != 0, bks = p.b_kstrided != 0;  // conv == 2: patch view, plain fills
    const bool kind_ok = (!ac && !bc) || (ac && !bc && !aks && !bks) || (!ac && bc && aks && bks);
    const bool nt_plain = kind_ok && dma_ok(p.A) && dma_ok(p.B);
    if (p.colsum && !(nt_plain && !no_glds && aks)) return CMDA_ERR_UNSUPPORTED;
    if (nt_plain && !no_glds) {
      // the 256x256 tile has two kernels: the ping-pong kernel (gemm_pp.hip: 32x32x16 MFMA, two wave groups alternating LDS reads and
      // MFMAs, bf16 epilogue image) for plain epilogues, gemm_glds_kernel for the rest (tile_hint bit 9: force the latter, tuning A/B)
      // Measured (profiles/r03_gemm_big.txt): the ping-pong kernel wins where the epilogue / short contraction dominates (K <= 1024:
      // 65536x1280x320 96 against 113 us, the head's pointwise data gradient 262144x1024x256 300 against 392 us) and on K-strided B
      // (8192^3 NN 863 against 778-837 TFLOP/s); long K-contiguous contractions and im2col views stay on gemm_glds_kernel (8192^3 NT
      // 1086 against 1024, the 3x3 bottleneck 1405 against 1776 us: the per-lane im2col address arithmetic sits in the LOAD segments)
      const bool pp_shape = (p.tile_hint > 0 && (p.tile_hint & 1024)) || (!ac && (p.K <= 1024 || bks));
      if (tile == 3 && pp_shape && !(p.tile_hint > 0 && (p.tile_hint & 512)) && !aks && !bc && !p.atomic && p.splits == 1 && !p.res &&
          !p.rowscale && p.beta == 0.f && !p.out_f32 && p.c_patch_ow == 0 && !p.colsum)
        return cmda_gemm_pp_(p, stream);
      if (tile == 3 && aks && bks && !ac && p.atomic && p.out_f32 && !(p.tile_hint > 0 && (p.tile_hint & 512)))
        return cmda_gemm_wg_(p, stream);   // weight-gradient form: gemm_wg.hip
      if (tile == 3) return cmda_gemm_glds_t3_(p, stream);
      if (tile == 0) return cmda_gemm_glds_t0_(p, stream);
      // the encoders' Linear layers / data gradients on the two small tiles: the lean instance (gemm_lean.hip), in the 4-stage latency
      // configuration where launch_glds would choose it (>= 12 k-tiles on a grid that is resident at once)
      if (!p.colstats && cmda_gemm_lean_ok_(p, tile)) {   // (fused column statistics: the general epilogue)
        const long tl = tile == 1 ? blocks(128, 64) : blocks(64, 64);
        const int force = p.tile_hint > 0 ? ((p.tile_hint >> 4) & 15) : 0;
        const bool four_stage = force ? force == 4 : (tl <= 256L * (tile == 1 ? 1 : 2) && nkt >= 12);
        return cmda_gemm_lean_(p, tile, four_stage ? 1 : 0, stream);
      }
      if (tile == 1) return cmda_gemm_glds_t1_(p, stream);
      return cmda_gemm_glds_t2_(p, stream);
    }
  }
  // the fused bias gradient lives in the LDS-DMA kernels only (bf16: above; split-bf16: the lean weight-gradient form)
  if (p.colsum && !(p.dtype == CMDA_F32X3 && cmda_gemm_x3_lean_ok_(p))) return CMDA_ERR_UNSUPPORTED;
  return cmda_gemm_reg_(p, tile, stream);
}

}  // namespace

extern "C" int cmda_gemm(const cmda_gemm_params_t* pp, void* stream) {
  if (!pp) return CMDA_ERR_SHAPE;
  GemmParams p = *pp;
  if (p.batch2 <= 0) p.batch2 = 1;
  if (p.M <= 0 || p.N <= 0 || p.batch <= 0) return CMDA_OK;
  if (p.K <= 0) return CMDA_ERR_SHAPE;
  if (p.atomic && !p.out_f32) return CMDA_ERR_UNSUPPORTED;
  if (p.splits > 1 && !p.atomic) return CMDA_ERR_UNSUPPORTED;
  if (p.atomic && (p.bias || p.act || p.res || p.rowscale)) return CMDA_ERR_UNSUPPORTED;
  if (p.colstats && (p.atomic || p.splits > 1 || p.act || p.rowscale || p.c_patch_ow > 0 || p.a_kstrided || p.batch != 1 || p.batch2 != 1 || (p.N & 3) ||
                     p.colstats_rows <= 0 || (p.colstats_rows & 255) || p.colsum))
    return CMDA_ERR_UNSUPPORTED;
  if (p.c_patch_ow > 0 && (p.atomic || p.res || p.batch != 1 || p.batch2 != 1 || p.c_patch_kh <= 0 || p.c_patch_kwci <= 0 ||
                           p.N != p.c_patch_kh * p.c_patch_kwci || p.M % p.c_patch_ow != 0 || (p.c_patch_kwci & 3)))
    return CMDA_ERR_UNSUPPORTED;
  if (p.dtype == CMDA_F32 || p.dtype == CMDA_F32X3) return launch_dtype<float>(p, stream);   // (fp32 storage: same tile heuristics)
  if (p.dtype == CMDA_BF16) return launch_dtype<bf16_t>(p, stream);
  return CMDA_ERR_DTYPE;
}

#ifdef CMDA_GEMM_TIMING
// the tuning build is ONE translation unit, so that every kernel stamps the same g_stamps array
#include "gemm_t0.hip"
#include "gemm_t1.hip"
#include "gemm_t2.hip"
#include "gemm_t3.hip"
#include "gemm_reg.hip"
#include "gemm_reg_f32_t0.hip"
#include "gemm_reg_f32_t1.hip"
#include "gemm_reg_f32_t2.hip"
#include "gemm_reg_bf16_t0.hip"
#include "gemm_reg_bf16_t1.hip"
#include "gemm_reg_bf16_t2.hip"
extern "C" int cmda_debug_gemm_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 256 * 8) == hipSuccess ? 0 : -3;
}
#endif
