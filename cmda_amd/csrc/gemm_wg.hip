// gemm_wg.hip -- 256x256 weight-gradient GEMM kernel: dW[m, n] += sum_k dY(k, m) * X(k, n), both operands K-STRIDED (rows = the
// contraction index: pixels / tokens), fp32 atomic accumulation, split-K over grid z.  The large weight gradients of the DAFormer
// head under autograd: the 3x3 bottleneck convolution (daformer_head.py:63-79: 256 x 9216 over 262144 pixels, X = im2col view of
// the NHWC input) and the pointwise convolutions of the separable ASPP branches (sep_aspp_head.py:18-27: 256 x 1024 over 262144).
//
// Why a kernel of its own: these contractions stream both operands from beyond L2 (~8 TB/s chip-wide into LDS), so the tile size
// IS the speed -- a 128x128 tile re-reads dY once per 128 columns and X once per 128 rows of dW (19 GB for the bottleneck, 2.9 ms),
// 256x256 halves both (9.7 GB).  gemm_glds_kernel's 8-wave tile spills in this operand mode (fragments of the 16x16x32 MFMA for a
// 64 x 128 wave tile: 12 x 4 VGPRs per 32-deep step + the im2col address state of 8 DMA pieces); on v_mfma_f32_32x32x16_bf16 the
// same wave tile needs 6 fragments per 16-deep step, and a 32x32 accumulator register is two 128-byte runs of one output row
// each -- the full-rate shape for global_atomic_add_f32 (MI355X_MICROARCH.md, Global float atomics).
//
// Structure: 512 threads = 8 waves as 4 (M) x 2 (N), wave tile 64 x 128 = 2 x 4 accumulators of 32x32 (128 VGPRs); k-tiles of 32
// pixels in FOUR LDS stages ([32 k][256] bf16 per operand, 16-byte slots XOR-swizzled by (k & 7) on the source address and on the
// read), filled by global_load_lds_dwordx4 (4 pieces of 1 KB per wave per k-tile), fragments by ds_read_b64_tr_b16; one barrier
// per k-tile that leaves the two younger tiles' DMA in flight (counted vmcnt) -- with one workgroup per CU the bytes in flight set
// the rate: two 64-deep stages (one tile in flight) ran the pointwise gradient at 3.1 TB/s of operand traffic.
#include "gemm_kernels.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16w;

#ifndef CMDA_EMU
static __device__ __forceinline__ f32x16w wg_mfma(u16x8 a, u16x8 b, f32x16w c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfx8, a), __builtin_bit_cast(bfx8, b), c, 0, 0, 0);
}
#else
static inline f32x16w wg_mfma(u16x8 a, u16x8 b, f32x16w c) {
  auto& w = emu::my_wave();
  const int l = emu::my_lane();
  for (int j = 0; j < 8; ++j) { w.fa[l][j] = bf2f(a[j]); w.fb[l][j] = bf2f(b[j]); }
  emu::wave_barrier();
  const int col = l & 31, h = l >> 5;
  f32x16w d = c;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    float acc = d[r];
    for (int hh = 0; hh < 2; ++hh)
      for (int j = 0; j < 8; ++j) acc = fmaf(w.fa[32 * hh + row][j], w.fb[32 * hh + col][j], acc);
    d[r] = acc;
  }
  emu::wave_barrier();
  return d;
}
#endif

constexpr int WG_T = 256, WG_NW = 8;

// fragment of a K-strided [32 k][256 c] tile for the 32x32x16 MFMA: lane l -> column c0 + (l & 31), k = 16 s + 8 (l >> 5) + j
static __device__ __forceinline__ u16x8 wg_frag(const bf16_t* tile, int c0, int s, int lane) {   // (any tile depth: lines of 256)
  const int gi = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  const int colbase = c0 + 16 * (gi & 1);
  const int cidx = (colbase >> 3) + (pp >> 1), half = (pp & 1) << 2;
  const int k0 = 16 * s + 8 * (gi >> 1) + q, k1 = k0 + 4;
  const u16x4 lo = lds_read_tr16(&tile[k0 * WG_T + ((cidx ^ (k0 & 7)) << 3) + half]);
  const u16x4 hi = lds_read_tr16(&tile[k1 * WG_T + ((cidx ^ (k1 & 7)) << 3) + half]);
  return u16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// (WG_BK, WG_NS) = (32, 4) for plain operands (three k-tiles of 32 KB in flight: the pointwise gradient 347 -> 296 us) and (64, 2)
// for the im2col view (its per-piece address arithmetic runs once per k-tile: the bottleneck 1787 us against 2163 at (32, 4))
template <bool BCONV, int WG_BK, int WG_NS, bool FAST = false>
static __device__ __forceinline__ void wg_body(const GemmParams& p, char* smem, int bt, int z) {
  typedef bf16_t T;
  constexpr int WG_SZ = WG_T * WG_BK;   // elements per operand per stage
  constexpr int PCS = WG_BK / 16;       // DMA pieces per wave per operand per k-tile
  T* const sAbase = reinterpret_cast<T*>(smem);
  T* const sBbase = sAbase + WG_NS * WG_SZ;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_n = (p.N + WG_T - 1) / WG_T;
  const long m0 = (long)(bt / tiles_n) * WG_T, n0 = (long)(bt % tiles_n) * WG_T;
  const int bz = z / p.splits, split = z - bz * p.splits;
  const int batch = bz / p.batch2, batch2 = bz - batch * p.batch2;
  const int nkt = (p.K + WG_BK - 1) / WG_BK;
  const int kt_per = (nkt + p.splits - 1) / p.splits;
  const int kt0 = split * kt_per, kt1 = min(nkt, kt0 + kt_per);
  if (kt0 >= kt1) return;
  const T* baseA = reinterpret_cast<const T*>(p.A.ptr) + (long)batch * p.A.batch_stride + (long)batch2 * p.A.batch2_stride;
  const T* baseB = reinterpret_cast<const T*>(p.B.ptr) + (long)batch * p.B.batch_stride + (long)batch2 * p.B.batch2_stride;

  DmaSrc<true, WG_T, false, WG_NW, 0, WG_BK, DmaMode<true, false, FAST>::value> dA;   // (FAST: gemm_kernels.h DmaSrc modes; a K-strided
  DmaSrc<true, WG_T, BCONV, WG_NW, 0, WG_BK, DmaMode<true, BCONV, FAST>::value> dB;   //  im2col B stays general)
  dA.init(p.A, baseA, wid, lane, m0, kt0);
  dB.init(p.B, baseB, wid, lane, n0, kt0);
  auto issue = [&](int stage, int kt) {
    char* la = reinterpret_cast<char*>(sAbase + stage * WG_SZ) + wid * PCS * 1024;
    char* lb = reinterpret_cast<char*>(sBbase + stage * WG_SZ) + wid * PCS * 1024;
    dA.cell_begin(kt, kt0, &p.A);
    dB.cell_begin(kt, kt0, &p.B);
#pragma unroll
    for (int j = 0; j < PCS; ++j) glds16(dA.get(p.A, baseA, j, kt, false), la + j * 1024);
#pragma unroll
    for (int j = 0; j < PCS; ++j) glds16(dB.get(p.B, baseB, j, kt, false), lb + j * 1024);
  };
  static_assert(decltype(dA)::J == PCS && decltype(dB)::J == PCS, "pieces per wave per k-tile");

  f32x16w acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // optional fused bias gradient (nn.Linear under autograd: db = column sums of dY): the workgroups of the first n-tile also add
  // up their A tile along k
  const bool do_colsum = p.colsum != nullptr && n0 == 0;
  float bsum = 0.f;

#pragma unroll
  for (int s = 0; s < WG_NS - 1; ++s)
    if (kt0 + s < kt1) issue(s, kt0 + s);
  int st = 0;
  for (int kt = kt0; kt < kt1; ++kt) {
    // k-tile kt has landed (the two younger tiles -- 4 pieces each -- may stay in flight; the tail issued fewer: drain);
    // past the barrier every wave is done reading stage st - 1, which the next DMA overwrites
    if (kt + WG_NS - 2 < kt1) pipe_barrier<(WG_NS - 2) * 2 * PCS>();
    else pipe_barrier<0>();
    {
      int sn = st + WG_NS - 1;
      if (sn >= WG_NS) sn -= WG_NS;
      if (kt + WG_NS - 1 < kt1) issue(sn, kt + WG_NS - 1);
    }
    const T* sA = sAbase + st * WG_SZ;
    const T* sB = sBbase + st * WG_SZ;
    if (do_colsum && tid < WG_T) {
#pragma unroll 8
      for (int k = 0; k < WG_BK; ++k) bsum += bf2f(sA[k * WG_T + (((tid >> 3) ^ (k & 7)) << 3) + (tid & 7)]);
    }
#pragma unroll
    for (int s = 0; s < WG_BK / 16; ++s) {
      u16x8 fa[2], fb[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = wg_frag(sA, wm * 64 + i * 32, s, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = wg_frag(sB, wn * 128 + j * 32, s, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = wg_mfma(fa[i], fb[j], acc[i][j]);
    }
    if (++st == WG_NS) st = 0;
  }

  if (do_colsum && tid < WG_T && m0 + tid < p.M) atomicAdd(p.colsum + m0 + tid, bsum);
  // acc[i][j][r]: m = m0 + wm*64 + i*32 + (r & 3) + 8*(r >> 2) + 4*(lane >> 5), n = n0 + wn*128 + j*32 + (lane & 31): one register
  // of one accumulator = two 128-byte runs (rows m and m + 4)
  const long cb = (long)batch * p.c_batch_stride + (long)batch2 * p.c_batch2_stride;
  float* C = reinterpret_cast<float*>(p.C) + cb;
  const int l31 = lane & 31, h = lane >> 5;
  const float alpha = p.alpha;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long n = n0 + wn * 128 + j * 32 + l31;
      if (n >= p.N) continue;
      const long cn = atomic_col(p, n);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < p.M) atomicAdd(C + m * p.ldc + cn, alpha * acc[i][j][r]);
      }
    }
}

template <bool BCONV, int WG_BK, int WG_NS, bool FAST = false>
__global__ __launch_bounds__(512, 1) void gemm_wg_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(1024))) char smem[(size_t)2 * WG_NS * WG_T * WG_BK * 2];
  // Workgroups are dealt round-robin to the 8 XCDs in launch order (tile fastest, then split).  The tiles of ONE split read the same K
  // slice of dY (and, for an im2col B, the same pixels under nine taps): dealt out as launched, every XCD streams every split's slice
  // through its own L2.  Give each XCD a contiguous range of (split, tile) pairs instead -- at most two splits per XCD for the
  // 36 x 7 grid of the 3x3 bottleneck.
  const unsigned nb = gridDim.x * gridDim.z, lin = blockIdx.x + gridDim.x * blockIdx.z;
  const unsigned q = nb / 8, r = nb % 8, xcd = lin % 8, loc = lin / 8;
  const unsigned logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  wg_body<BCONV, WG_BK, WG_NS, FAST>(p, smem, (int)(logical % gridDim.x), (int)(logical / gridDim.x));
}

// grouped form (gemm_grouped.hip: the deferred weight gradients of an encoder stage whose outputs are at least 256 x 256 -- MiT
// stages 3 / 4: 320 ... 2048 channels): `tab` / `blk` as for gemm_glds_grouped_kernel, block index inside a problem = z * tiles + tile
template <bool BCONV, int WG_BK, int WG_NS, bool FAST = false>
__global__ __launch_bounds__(512, 1) void gemm_wg_grouped_kernel(const GemmParams* __restrict__ tab, const int* __restrict__ blk) {
  __shared__ __attribute__((aligned(1024))) char smem[(size_t)2 * WG_NS * WG_T * WG_BK * 2];
#ifndef CMDA_EMU
  const int prob = __builtin_amdgcn_readfirstlane(blk[2 * blockIdx.x]), loc = __builtin_amdgcn_readfirstlane(blk[2 * blockIdx.x + 1]);
#else
  const int prob = blk[2 * blockIdx.x], loc = blk[2 * blockIdx.x + 1];
#endif
  if (prob < 0) return;
  const GemmParams& p = tab[prob];
  const int ntile = (int)((p.M + WG_T - 1) / WG_T) * ((p.N + WG_T - 1) / WG_T);
  const int z = loc / ntile;
  wg_body<BCONV, WG_BK, WG_NS, FAST>(p, smem, loc - z * ntile, z);
}

}  // namespace

int cmda_gemm_wg_grouped_(const cmda_gemm_params_t* tab, const void* blk, int nblocks, int bconv, void* stream) {
  if (nblocks <= 0) return CMDA_OK;
  const dim3 grid((unsigned)nblocks), b(512);
  const int* bp = reinterpret_cast<const int*>(blk);
  // bconv = operand class of the bucket (gemm_grouped.hip): 0 plain with K % 64 == 0 (running-pointer DMA sources), 1 general, 2 im2col B
  if (bconv == 2) CMDA_LAUNCH((gemm_wg_grouped_kernel<true, 64, 2, false>), grid, b, 0, stream, tab, bp);
  else if (bconv == 1) CMDA_LAUNCH((gemm_wg_grouped_kernel<false, 32, 4, false>), grid, b, 0, stream, tab, bp);
  else CMDA_LAUNCH((gemm_wg_grouped_kernel<false, 32, 4, true>), grid, b, 0, stream, tab, bp);
  CMDA_CHECK_LAUNCH();
}

// cross-unit entry (gemm.hip decides eligibility and the split count)
int cmda_gemm_wg_(const cmda_gemm_params_t& p, void* stream) {
  const long tiles = (long)((p.M + WG_T - 1) / WG_T) * ((p.N + WG_T - 1) / WG_T);
  const long zz = (long)p.batch * p.batch2 * p.splits;
  if (tiles > 0x7fffffffL || zz > 65535) return CMDA_ERR_SHAPE;
  const dim3 grid((unsigned)tiles, 1, (unsigned)zz), blk(512);
  const bool nofast = p.tile_hint > 0 && (p.tile_hint & 2048);
  if (p.B.conv == 1) {
    // FAST: dY in the running-pointer mode, the im2col operand in the row-fast mode (stride-1 "same" convolution, output rows of
    // whole k-tiles: the head's 3 x 3 bottleneck).  tile_hint bit 12 (tuning A/B): the plain operands' four half-depth stages for the
    // im2col view too -- measured SLOWER (1907 against 1690 us on the general path's two full-depth stages, gpurun r04f)
    typedef DmaSrc<true, WG_T, false, WG_NW, 0, 64, 1> FA;
    typedef DmaSrc<true, WG_T, true, WG_NW, 0, 64, 3> FB;
    typedef DmaSrc<true, WG_T, false, WG_NW, 0, 32, 1> FA32;
    typedef DmaSrc<true, WG_T, true, WG_NW, 0, 32, 3> FB32;
    if (!nofast && FA32::mode_ok(p.A, 1) && FB32::mode_ok(p.B, 3) && (p.tile_hint > 0 && (p.tile_hint & 4096)))
      CMDA_LAUNCH((gemm_wg_kernel<true, 32, 4, true>), grid, blk, 0, stream, p);
    else if (!nofast && FA::mode_ok(p.A, 1) && FB::mode_ok(p.B, 3)) CMDA_LAUNCH((gemm_wg_kernel<true, 64, 2, true>), grid, blk, 0, stream, p);
    else CMDA_LAUNCH((gemm_wg_kernel<true, 64, 2, false>), grid, blk, 0, stream, p);
  } else {
    typedef DmaSrc<true, WG_T, false, WG_NW, 0, 32, 1> FA;
    if (!nofast && FA::mode_ok(p.A, 1) && FA::mode_ok(p.B, 1)) CMDA_LAUNCH((gemm_wg_kernel<false, 32, 4, true>), grid, blk, 0, stream, p);
    else CMDA_LAUNCH((gemm_wg_kernel<false, 32, 4, false>), grid, blk, 0, stream, p);
  }
  CMDA_CHECK_LAUNCH();
}
