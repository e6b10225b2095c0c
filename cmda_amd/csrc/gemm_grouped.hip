// gemm_grouped.hip -- planning and dispatch of GROUPED GEMM launches (cmda_gemm_grouped): many weight-gradient GEMMs of a
// backward pass in ONE grid per (tile, operand mode).
//
// Why: at the UDA step's 2 + 2 samples per GPU an encoder's ~310 weight gradients (mix_transformer.py Linear / Conv2d layers under
// autograd: dW += dY^T X) are launches of 100-200 blocks each that cannot fill 256 CUs and cost ~20 us apiece (12.9 of the 39 ms of
// kernel time of the encoder backward, profiles/r03_*): nothing depends on them until the optimizer step, so the backward pass
// only QUEUES them (ops.gemm_deferral) and hands the whole list to this entry point once per encoder stage.
//
// Layout of the workspace (same bytes on the host and on the device): [n parameter blocks, splits resolved][pad to 16]
// [int32 pairs {problem, block-in-problem}, bucket after bucket].  Problems the grouped kernels do not cover (fp32 parity mode,
// operands off the LDS-DMA path, non-atomic outputs) are launched one by one through cmda_gemm, in list order.
#include <vector>

#include "gemm_kernels.h"

namespace {

struct Plan {
  int bucket;       // 0..3 = (tile 64x64 | 128x128) x (B plain / patch | B im2col view); -1 = single launch
  int splits;
  long blocks;
};

static bool dma_view_ok(const GemmView& v) {
  return v.vec_ok && v.in_dil <= 1 && v.R < (1L << 31) && v.Cc < (1L << 31) && (!v.conv || (v.H < 32768 && v.W < 32768)) &&
         (v.conv || (v.ld % 8) == 0) && (v.Cc % 8) == 0 &&
         (v.conv != 2 || ((v.KW * v.C) % 64 == 0 && v.KH == v.stride && v.KW == v.stride && v.pad == 0 && v.dil == 1 &&
                          v.H == v.OH * v.stride && v.W == v.OW * v.stride));
}

// kt_target: k-tiles (64 deep) one block should run: bounds a block's time (load balance inside the big grid) while keeping the
// atomic traffic (one fp32 atomic per output element per split) low
static Plan plan_one(const GemmParams& p, int kt_target) {
  Plan pl{-1, 1, 0};
  const bool ok = p.dtype == CMDA_BF16 && p.atomic && p.out_f32 && p.a_kstrided && p.b_kstrided && p.A.conv != 1 && p.batch >= 1 &&
                  p.M > 0 && p.N > 0 && p.K > 0 && dma_view_ok(p.A) && dma_view_ok(p.B) && p.tile_hint == 0 && p.splits <= 0 &&
                  !p.bias && !p.act && !p.res && !p.rowscale && p.c_patch_ow == 0;
  if (!ok) return pl;
  const int b2 = p.batch2 > 0 ? p.batch2 : 1;
  const bool big = (p.M % 128 == 0) && (p.N % 128 == 0);
  const int bm = big ? 128 : 64;
  const long tiles = (long)((p.M + bm - 1) / bm) * ((p.N + bm - 1) / bm);
  const int nkt = (p.K + 63) / 64;
  long s = (nkt + kt_target - 1) / kt_target;
  s = std::max<long>(1, std::min<long>(s, 256));
  pl.bucket = (big ? 2 : 0) + (p.B.conv == 1 ? 1 : 0);
  pl.splits = (int)s;
  pl.blocks = tiles * s * p.batch * b2;
  return pl;
}

__global__ void upload_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long n16) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}

static long plan_all(const GemmParams* params, int n, std::vector<Plan>& plans) {
  plans.resize(n);
  long total = 0;
  // fewer k-tiles per block while the whole group would not fill the chip a few times over
  for (int kt_target : {48, 24, 12, 8}) {
    total = 0;
    for (int i = 0; i < n; ++i) {
      plans[i] = plan_one(params[i], kt_target);
      if (plans[i].bucket >= 0) total += plans[i].blocks;
    }
    if (total >= 4096) break;
  }
  return total;
}

}  // namespace

extern "C" int64_t cmda_gemm_grouped_ws_bytes(const cmda_gemm_params_t* params, int n) {
  if (!params || n <= 0) return 0;
  std::vector<Plan> plans;
  const long total = plan_all(params, n, plans);
  const long tab = ((long)n * (long)sizeof(GemmParams) + 15) / 16 * 16;
  return tab + (total * 8 + 15) / 16 * 16;
}

extern "C" int cmda_gemm_grouped(const cmda_gemm_params_t* params, int n, void* ws_host, void* ws_dev, int64_t ws_bytes, int upload,
                                 void* stream) {
  if (n <= 0) return CMDA_OK;
  if (!params) return CMDA_ERR_SHAPE;
  std::vector<Plan> plans;
  const long total = plan_all(params, n, plans);
  const long tab_bytes = ((long)n * (long)sizeof(GemmParams) + 15) / 16 * 16;
  const long need = tab_bytes + (total * 8 + 15) / 16 * 16;
  long start[4] = {0, 0, 0, 0}, count[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i)
    if (plans[i].bucket >= 0) count[plans[i].bucket] += plans[i].blocks;
  for (int b = 1; b < 4; ++b) start[b] = start[b - 1] + count[b - 1];
  if (total > 0) {
    if (!ws_host || !ws_dev || ws_bytes < need) return CMDA_ERR_SHAPE;
    if (total > 0x7fffffffL) return CMDA_ERR_SHAPE;
    if (upload) {
      GemmParams* tab = reinterpret_cast<GemmParams*>(ws_host);
      int* blk = reinterpret_cast<int*>(reinterpret_cast<char*>(ws_host) + tab_bytes);
      long fill[4] = {start[0], start[1], start[2], start[3]};
      for (int i = 0; i < n; ++i) {
        tab[i] = params[i];
        if (tab[i].batch2 <= 0) tab[i].batch2 = 1;
        if (plans[i].bucket < 0) continue;
        tab[i].splits = plans[i].splits;
        long& f = fill[plans[i].bucket];
        for (long k = 0; k < plans[i].blocks; ++k, ++f) {
          blk[2 * f] = i;
          blk[2 * f + 1] = (int)k;
        }
      }
      const long n16 = need / 16;
      const int grid = (int)std::min<long>((n16 + 255) / 256, 512);
      CMDA_LAUNCH(upload_kernel, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const uint4*>(ws_host), reinterpret_cast<uint4*>(ws_dev), n16);
    }
    const GemmParams* dtab = reinterpret_cast<const GemmParams*>(ws_dev);
    const char* dblk = reinterpret_cast<const char*>(ws_dev) + tab_bytes;
    for (int b = 0; b < 4; ++b) {
      if (!count[b]) continue;
      const void* bp = dblk + start[b] * 8;
      const int rc = (b & 2) ? cmda_gemm_grouped_t0_(dtab, bp, (int)count[b], b & 1, stream) : cmda_gemm_grouped_t2_(dtab, bp, (int)count[b], b & 1, stream);
      if (rc != CMDA_OK) return rc;
    }
  }
  for (int i = 0; i < n; ++i)
    if (plans[i].bucket < 0) {
      const int rc = cmda_gemm(&params[i], stream);
      if (rc != CMDA_OK) return rc;
    }
  return CMDA_OK;
}
