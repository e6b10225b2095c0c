// gemm_t4.hip -- the 64 x 320 ROW-PANEL instantiation of the LDS-DMA GEMM kernel: one workgroup (8 waves, 16 x 160 each) owns 64 full
// rows of a 320-column output (or one 320-column slice of a wider one), two 48-KiB LDS stages.  Built for the Linear layers of the MiT
// stage with C = 320 (q / proj 320 x 320, kv 640 x 320, fc1 1280 x 320, fc2 320 x 1280 -- mix_transformer.py:31-44,62-66,80-102) as
// the carrier of a LayerNorm epilogue (whole rows per workgroup: VERDICT r03 item 2 ii).  MEASURED AND NOT SELECTED by the heuristics
// (tile_hint 5 only: tuning sweeps and tests): graph-timed on MI355X (tools/dbg/rp_bench.py, gpurun r04d) 8192 x 320 x 320 takes 12.1 us
// against 7.9 on the 64 x 64 tile, 8192 x 320 x 1280 29.4 against 17.7 -- per k-tile every wave re-reads 22 KiB of B fragments from LDS
// for 20 MFMAs, and the 4-wave form (32 x 160 per wave) serialises its fragment reads behind its MFMAs (15.0 / 35.8 us); the 4-12 us
// lost exceed the ~7.8 us LayerNorm launch the epilogue would remove.  Templates: gemm_kernels.h.
#include "gemm_kernels.h"

int cmda_gemm_glds_t4_(const cmda_gemm_params_t& p, void* stream) {
  constexpr int BM = 64, BN = 320;
  // K-strided tiles need TILE / 8 chunks per line to divide a wave's 64 lanes (DmaSrc): 320 does not -- plain K-contiguous operands only
  if (p.a_kstrided || p.b_kstrided || p.A.conv || p.B.conv) return CMDA_ERR_UNSUPPORTED;
  const long tiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const long zz = (long)p.batch * p.batch2 * p.splits;
  if (tiles > 0x7fffffffL || zz > 65535) return CMDA_ERR_SHAPE;
  dim3 grid((unsigned)tiles, 1, (unsigned)zz);
  return launch_glds_mode<1, 10, 8, 2, false, false, false, false>(p, grid, stream);   // 8 waves (4 x 2), 16 x 160 per wave
}
