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

// tile kinds of the grouped kernels: {BM, BN, resident blocks per CU (LDS 2 stages x (BM + BN) x 128 B)}
// kind 4 = gemm_wg.hip's 256x256 tile (8 waves, 32x32x16 MFMA): outputs of at least 256 x 256 -- both operands of these contractions
// stream from beyond L2, so operand bytes per FLOP decide
// kind 5 = the lean split-bf16 kernel's weight-gradient form (gemm_x3_lean.hip: CMDA_F32X3, 64 x 64 tiles on eight waves, 32-deep k-tiles,
// two workgroups per CU)
constexpr int kTileBM[6] = {64, 128, 64, 128, 256, 64}, kTileBN[6] = {64, 64, 128, 128, 256, 64}, kTileRes[6] = {4, 3, 3, 2, 1, 2};
// bucket = tile kind * 3 + operand class: 0 = both operands plain with K a multiple of 64 (the kernels' running-pointer DMA sources,
// gemm_kernels.h DmaSrc mode 1), 1 = plain / patch views in general, 2 = B is an im2col view
constexpr int kBuckets = 18;   // (kind 5 uses its first bucket only)

struct Plan {
  int bucket;       // -1 = single launch through cmda_gemm
  int splits;
  int tiles;        // output tiles of one (batch, split)
  int nkt;          // 64-deep k-tiles of the whole contraction
  long blocks;      // tiles * splits * batch * batch2
};

static bool dma_view_ok(const GemmView& v) {
  return v.vec_ok && v.in_dil <= 1 && v.R < (1L << 31) && v.Cc < (1L << 31) && (!v.conv || (v.H < 32768 && v.W < 32768)) &&
         (v.conv || (v.ld % 8) == 0) && (v.Cc % 8) == 0 &&
         (v.conv != 2 || ((v.KW * v.C) % 64 == 0 && v.KH == v.stride && v.KW == v.stride && v.pad == 0 && v.dil == 1 &&
                          v.H == v.OH * v.stride && v.W == v.OW * v.stride));
}

// which problems the grouped kernels take, and on which tile: the largest tile that divides the output evenly (a 64x64 tile moves
// 16 KB through L2 -> LDS per 0.5 MFLOP, 128x128 32 KB per 2.1 MFLOP: the k-loop is bound by that per-CU rate)
static Plan classify(const GemmParams& p) {
  Plan pl{-1, 1, 0, 0, 0};
  if (p.dtype == CMDA_F32X3) {   // split-bf16 mode: the weight gradients the lean split kernel takes (its own eligibility test)
    const int b2 = p.batch2 > 0 ? p.batch2 : 1;
    if (!(p.atomic && p.out_f32 && p.a_kstrided && p.b_kstrided && p.tile_hint == 0 && p.splits <= 0 && p.batch == 1 && b2 == 1 && p.M > 0 && p.N > 0 &&
          cmda_gemm_x3_lean_ok_(p)) || 2.0 * p.M * p.N * (double)p.K > 30e9)
      return pl;
    pl.bucket = 5 * 3;
    pl.tiles = ((p.M + 63) / 64) * ((p.N + 63) / 64);
    pl.nkt = p.K / 32;
    return pl;
  }
  const bool ok = p.dtype == CMDA_BF16 && p.atomic && p.out_f32 && p.a_kstrided && p.b_kstrided && p.A.conv != 1 && p.batch >= 1 &&
                  p.M > 0 && p.N > 0 && p.K > 0 && dma_view_ok(p.A) && dma_view_ok(p.B) && p.tile_hint == 0 && p.splits <= 0 &&
                  !p.bias && !p.act && !p.res && !p.rowscale && p.c_patch_ow == 0;
  if (!ok) return pl;
  // a problem that fills the chip by itself (the decode head's weight gradients over 262144 rows: 0.14 - 1.2 TFLOP each) keeps
  // its own launch and the single-launch heuristics (256x256 / 128x128 tiles, wave-quantised split-K): grouped, its >= 144 tiles
  // per split no longer run side by side on one XCD and re-stream their K slice (measured 5.3 against 2.9 ms)
  const int b2c = p.batch2 > 0 ? p.batch2 : 1;
  if (2.0 * p.M * p.N * (double)p.K * p.batch * b2c > 30e9) return pl;
  // (only when the zero padding of the last tiles stays under 10 %: MiT's 320-channel stage padded to 512 lost more MFMA time on
  // this kernel than the operand traffic it saved -- 6.2 ms per step against 5.1 on the small tiles)
  const long pm = (p.M + 255) / 256 * 256, pn = (p.N + 255) / 256 * 256;
  const bool big = p.M >= 256 && p.N >= 256 && (double)pm * pn <= 1.1 * (double)p.M * p.N;
  // (128-wide tiles over the 320-channel dimensions with a half-empty last tile -- 20 % padding for 20 - 28 % fewer operand bytes --
  // measured slower as well: the step 58.0 - 58.1 -> 58.2 - 58.4 ms, gpurun r04pad)
  const int kind = big ? 4 : (p.M % 128 == 0 ? 1 : 0) + (p.N % 128 == 0 ? 2 : 0);
  // the running-pointer ("fast") bucket: eligibility with the k-tile depth of the kernel that will run it -- kind 4 is
  // gemm_wg_grouped_kernel's 32-deep k-tile, and a K-strided PATCH view whose rows are exactly one 64-deep k-tile (OW == 64) passes
  // the 64-deep test while its per-lane `ow` does change between 32-deep tiles (wrong weight gradients, latent)
  typedef DmaSrc<true, 64, false, 4, 0, 64, 1> FK64;   // (tile size irrelevant for the eligibility test)
  typedef DmaSrc<true, 64, false, 4, 0, 32, 1> FK32;
  const bool fast = kind == 4 ? (FK32::mode_ok(p.A, 1) && FK32::mode_ok(p.B, 1)) : (FK64::mode_ok(p.A, 1) && FK64::mode_ok(p.B, 1));
  pl.bucket = kind * 3 + (p.B.conv == 1 ? 2 : fast ? 0 : 1);
  pl.tiles = ((p.M + kTileBM[kind] - 1) / kTileBM[kind]) * ((p.N + kTileBN[kind] - 1) / kTileBN[kind]);
  pl.nkt = (p.K + 63) / 64;
  return pl;
}

__global__ void upload_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long n16) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}

struct Unit { int prob, z, tiles, len; };   // one (batch, split) of a problem: `tiles` workgroups running `len` k-tiles each

// Split choice + block map of one bucket.  A unit's workgroups read the same K slice of both operands, so a unit goes to ONE XCD
// (workgroups b, b + 8, ... of the grid): there its tiles run side by side and share the slice through that XCD's L2 -- dealt
// round-robin over the XCDs instead, every XCD streams every operand (measured: 18 GB through the memory side for 1.9 GB of
// operands on an encoder stage's group).  Units are placed longest-first on the least loaded XCD (LPT), and run longest-first
// inside an XCD.  Splits: a block should run ~1/3 of the bucket's ideal time at most (tail), never fewer than 16 k-tiles (one
// fp32 atomic per output element per split: ~1.3 TB/s chip-wide).
struct BucketPlan {
  std::vector<Unit> xcd[8];
  long rows = 0;   // workgroups per XCD after padding: the grid is 8 * rows
};

static void plan_bucket(const GemmParams* params, std::vector<Plan>& plans, const std::vector<int>& members, int kind, BucketPlan& bp) {
  double work = 0;
  for (int i : members) {
    const GemmParams& p = params[i];
    const int b2 = p.batch2 > 0 ? p.batch2 : 1;
    work += (double)plans[i].tiles * plans[i].nkt * p.batch * b2;
  }
  const double slots = 256.0 * kTileRes[kind];
  const int len_target = (int)std::max(16.0, std::min(4096.0, work / slots / 3.0));
  std::vector<Unit> units;
  for (int i : members) {
    const GemmParams& p = params[i];
    const int b2 = p.batch2 > 0 ? p.batch2 : 1;
    Plan& pl = plans[i];
    int s = (pl.nkt + len_target - 1) / len_target;
    s = std::max(1, std::min(s, std::max(1, pl.nkt / 8)));
    s = std::min(s, 4096);
    pl.splits = s;
    pl.blocks = (long)pl.tiles * s * p.batch * b2;
    const int per = (pl.nkt + s - 1) / s;
    for (int z = 0; z < s * p.batch * b2; ++z) units.push_back(Unit{i, z, pl.tiles, per});
  }
  std::stable_sort(units.begin(), units.end(), [](const Unit& a, const Unit& b) { return (long)a.len * a.tiles > (long)b.len * b.tiles; });
  double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (const Unit& u : units) {
    int best = 0;
    for (int x = 1; x < 8; ++x)
      if (load[x] < load[best]) best = x;
    bp.xcd[best].push_back(u);
    load[best] += (double)u.len * u.tiles;
  }
  bp.rows = 0;
  for (int x = 0; x < 8; ++x) {
    std::stable_sort(bp.xcd[x].begin(), bp.xcd[x].end(), [](const Unit& a, const Unit& b) { return a.len > b.len; });
    long r = 0;
    for (const Unit& u : bp.xcd[x]) r += u.tiles;
    bp.rows = std::max(bp.rows, r);
  }
}

struct GroupPlan {
  std::vector<Plan> plans;
  BucketPlan buckets[kBuckets];
  long start[kBuckets], total = 0;   // first entry of each bucket in the block map, entries overall
};

static void plan_all(const GemmParams* params, int n, GroupPlan& g) {
  g.plans.resize(n);
  std::vector<int> members[kBuckets];
  for (int i = 0; i < n; ++i) {
    g.plans[i] = classify(params[i]);
    if (g.plans[i].bucket >= 0) members[g.plans[i].bucket].push_back(i);
  }
  g.total = 0;
  for (int b = 0; b < kBuckets; ++b) {
    g.start[b] = g.total;
    if (members[b].empty()) continue;
    plan_bucket(params, g.plans, members[b], b / 3, g.buckets[b]);
    g.total += g.buckets[b].rows * 8;
  }
}

}  // namespace

extern "C" int64_t cmda_gemm_grouped_ws_bytes(const cmda_gemm_params_t* params, int n) {
  if (!params || n <= 0) return 0;
  GroupPlan g;
  plan_all(params, n, g);
  const long tab = ((long)n * (long)sizeof(GemmParams) + 15) / 16 * 16;
  return tab + (g.total * 8 + 15) / 16 * 16;
}

extern "C" int cmda_gemm_grouped(const cmda_gemm_params_t* params, int n, void* ws_host, void* ws_dev, int64_t ws_bytes, int upload,
                                 void* stream) {
  if (n <= 0) return CMDA_OK;
  if (!params) return CMDA_ERR_SHAPE;
  GroupPlan g;
  plan_all(params, n, g);
  const long tab_bytes = ((long)n * (long)sizeof(GemmParams) + 15) / 16 * 16;
  const long need = tab_bytes + (g.total * 8 + 15) / 16 * 16;
  if (g.total > 0) {
    if (!ws_host || !ws_dev || ws_bytes < need) return CMDA_ERR_SHAPE;
    if (g.total > 0x7fffffffL) return CMDA_ERR_SHAPE;
    if (upload) {
      GemmParams* tab = reinterpret_cast<GemmParams*>(ws_host);
      int* blk = reinterpret_cast<int*>(reinterpret_cast<char*>(ws_host) + tab_bytes);
      for (int i = 0; i < n; ++i) {
        tab[i] = params[i];
        if (tab[i].batch2 <= 0) tab[i].batch2 = 1;
        if (g.plans[i].bucket >= 0) tab[i].splits = g.plans[i].splits;
      }
      for (int b = 0; b < kBuckets; ++b) {
        const BucketPlan& bp = g.buckets[b];
        if (!bp.rows) continue;
        int* base = blk + 2 * g.start[b];
        for (int x = 0; x < 8; ++x) {
          long r = 0;
          for (const Unit& u : bp.xcd[x])
            for (int t = 0; t < u.tiles; ++t, ++r) {
              base[2 * (r * 8 + x)] = u.prob;
              base[2 * (r * 8 + x) + 1] = u.z * u.tiles + t;
            }
          for (; r < bp.rows; ++r) {
            base[2 * (r * 8 + x)] = -1;
            base[2 * (r * 8 + x) + 1] = 0;
          }
        }
      }
      const long n16 = need / 16;
      const int grid = (int)std::min<long>((n16 + 255) / 256, 512);
      CMDA_LAUNCH(upload_kernel, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const uint4*>(ws_host), reinterpret_cast<uint4*>(ws_dev), n16);
    }
    const GemmParams* dtab = reinterpret_cast<const GemmParams*>(ws_dev);
    const char* dblk = reinterpret_cast<const char*>(ws_dev) + tab_bytes;
    for (int b = 0; b < kBuckets; ++b) {
      const long cnt = g.buckets[b].rows * 8;
      if (!cnt) continue;
      const void* bptr = dblk + g.start[b] * 8;
      const int kind = b / 3, bc = b % 3;   // bc: operand class (0 fast plain, 1 general, 2 im2col B)
      const int rc = kind == 0 ? cmda_gemm_grouped_t2_(dtab, bptr, (int)cnt, bc, stream)
                   : kind == 1 ? cmda_gemm_grouped_t1_(dtab, bptr, (int)cnt, bc, stream)
                   : kind == 2 ? cmda_gemm_grouped_t3_(dtab, bptr, (int)cnt, bc, stream)
                   : kind == 3 ? cmda_gemm_grouped_t0_(dtab, bptr, (int)cnt, bc, stream)
                   : kind == 4 ? cmda_gemm_wg_grouped_(dtab, bptr, (int)cnt, bc, stream)
                               : cmda_gemm_x3_lean_grouped_(dtab, bptr, (int)cnt, stream);
      if (rc != CMDA_OK) return rc;
    }
  }
  for (int i = 0; i < n; ++i)
    if (g.plans[i].bucket < 0) {
      const int rc = cmda_gemm(&params[i], stream);
      if (rc != CMDA_OK) return rc;
    }
  return CMDA_OK;
}
