// gemm_t4.hip -- the 64 x 320 ROW-PANEL instantiation of the LDS-DMA GEMM kernel: one workgroup owns 64 full rows of a 320-column
// output (or one 320-column slice of a wider one), three 48-KiB LDS stages.  For the Linear layers of the MiT stage with C = 320 (40
// of MiT-B5's 52 blocks: q / proj 320 x 320, kv 640 x 320, fc1 1280 x 320, fc2 320 x 1280 -- mix_transformer.py:31-44,62-66,80-102)
// at the UDA step's 2 + 2 samples (M = 2048 ... 8192 rows): a lone 64 x 64 block keeps one 16-KiB k-tile in flight against ~1 us of
// load latency (~15 B/clk per CU), the row panel keeps 96 KiB in flight, reads the weights ONCE per 64 rows instead of once per tile
// row, and its workgroups hold whole rows -- what a LayerNorm epilogue needs.  Templates: gemm_kernels.h; chosen by launch_dtype.
#include "gemm_kernels.h"

int cmda_gemm_glds_t4_(const cmda_gemm_params_t& p, void* stream) {
  constexpr int BM = 64, BN = 320;
  // K-strided tiles need TILE / 8 chunks per line to divide a wave's 64 lanes (DmaSrc): 320 does not -- plain K-contiguous operands only
  if (p.a_kstrided || p.b_kstrided || p.A.conv || p.B.conv) return CMDA_ERR_UNSUPPORTED;
  const long tiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const long zz = (long)p.batch * p.batch2 * p.splits;
  if (tiles > 0x7fffffffL || zz > 65535) return CMDA_ERR_SHAPE;
  dim3 grid((unsigned)tiles, 1, (unsigned)zz);
  const int var = p.tile_hint > 0 ? ((p.tile_hint >> 4) & 15) : 0;   // tuning: wave count / stage variants
  if (var == 1) return launch_glds_ns<1, 10, 8, 3>(p, grid, stream);   // 8 waves (4 x 2), 16 x 160 per wave
  if (var == 2) return launch_glds_ns<1, 10, 8, 2>(p, grid, stream);
  if (var == 3) return launch_glds_ns<2, 10, 4, 2>(p, grid, stream);
  return launch_glds_ns<2, 10, 4, 3>(p, grid, stream);
}
