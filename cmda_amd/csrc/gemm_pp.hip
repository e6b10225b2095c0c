// gemm_pp.hip -- the 256x256 "ping-pong" GEMM kernel for the large contractions of the CMDA step: the DAFormer head's 3x3
// bottleneck convolution forward / data gradient (daformer_head.py:63-79: 262144 x 256 x 9216 and 262144 x 1024 x 2304 at the
// UDA step's 16 decoder images), its pointwise convolutions and their data gradients (sep_aspp_head.py:18-27), the CycleGAN
// generator's 3x3 convolutions -- and the square benchmark shapes (tools/gemm_bench.py --big).
//
// C[m,n] = act(alpha * sum_k A(m,k) B(n,k) + bias[n]), bf16 in / bf16 out, fp32 accumulate.
//   A: K-contiguous view, plain or im2col (conv = 1 / patch view conv = 2);  B: plain, K-contiguous [N][K] or K-strided [K][N].
// Everything else (residual / row-scale / beta epilogues, fp32 or atomic outputs, K-strided A) stays on gemm_glds_kernel.
//
// gfx950 structure (MI355X_MICROARCH.md "Two waves per SIMD", cdna_hip_programming.md 5 "8-phase template"):
//   * 512 threads = 8 waves as 2 (M) x 4 (N); a wave owns 128 x 64 of the tile as 4 x 2 accumulators of
//     v_mfma_f32_32x32x16_bf16 (128 VGPRs).  The two waves that share a SIMD belong to different M halves (groups G0 / G1).
//   * k-tiles of 64 in two LDS stages (2 x 64 KB), filled by global_load_lds_dwordx4 (8 x 1 KB pieces per wave per k-tile),
//     16-byte slots XOR-swizzled by (line & 7) on the SOURCE address and on the read.
//   * every k-tile is 4 phases (one 64 x 32 quadrant of the wave tile x K = 64 = 8 MFMAs); a phase is a LOAD segment
//     (ds_read_b128 of the quadrant's fragments + a few DMA pieces of the NEXT k-tile) and a COMPUTE segment (the 8 MFMAs),
//     each closed by a raw s_barrier.  G1 runs ONE barrier behind G0, so on every SIMD one wave issues MFMAs while its
//     partner reads LDS / issues DMA: the matrix pipe never waits for the loads of its own wave.
//   * the next tile's pieces go out in phases 0-2 (3 + 3 + 2), each wave retires its own with vmcnt(0) in phase 3; the first
//     read of that stage is two barriers later for either group (reads of a DMA-filled buffer are ordered by the issuing
//     wave's vmcnt + a barrier the reader has passed, nothing else).
//   * epilogue: operands are fed to the MFMA swapped (B fragment first), so a lane holds 4 consecutive n of one row m:
//     bias / activation in registers, packed to bf16, 8-byte LDS stores into a padded [256][264] bf16 image, then every
//     thread streams 16-byte pieces of whole rows to HBM (2 rows per wave instruction).
#include "gemm_kernels.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

#ifndef CMDA_EMU
static __device__ __forceinline__ f32x16 mfma_bf16_32x32x16(u16x8 a, u16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfx8, a), __builtin_bit_cast(bfx8, b), c, 0, 0, 0);
}
static __device__ __forceinline__ void raw_barrier() { __builtin_amdgcn_s_barrier(); }
static __device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
static __device__ __forceinline__ void prio(int hi) {
  if (hi) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}
static __device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
static __device__ __forceinline__ void wait_dma_lds() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
#else
// 32x32x16: lane l holds A[row l&31][k = 8*(l>>5)+j], B[k = 8*(l>>5)+j][col l&31]; D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)
static inline f32x16 mfma_bf16_32x32x16(u16x8 a, u16x8 b, f32x16 c) {
  auto& w = emu::my_wave();
  const int l = emu::my_lane();
  for (int j = 0; j < 8; ++j) { w.fa[l][j] = bf2f(a[j]); w.fb[l][j] = bf2f(b[j]); }
  emu::wave_barrier();
  const int col = l & 31, h = l >> 5;
  f32x16 d = c;
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
static inline void raw_barrier() { __syncthreads(); }
static inline void sched_fence() {}
static inline void prio(int) {}
static inline void wait_lds() {}
static inline void wait_dma_lds() { emu::wave_barrier(); }
#endif

// -DCMDA_PP_TIMING (tuning builds, tools/dbg/pp_phase.py): lane 0 of waves 0 (G0) and 4 (G1) of workgroup 7 stamps s_memtime at the
// segment boundaries of k-tiles 8..11 into g_pp_stamps[wave group][k-tile][q][0 = LOAD start, 1 = barrier passed, 2 = MFMAs issued,
// 3 = closing barrier passed]
#ifdef CMDA_PP_TIMING
__device__ unsigned long long g_pp_stamps[2 * 4 * 4 * 4];
#define PP_STAMP(i) do { if (lane == 0 && (wid & 3) == 0 && blockIdx.x == 7 && kt >= 8 && kt < 12) \
    g_pp_stamps[((grp * 4 + (kt - 8)) * 4 + q) * 4 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PP_STAMP(i) do { } while (0)
#endif

constexpr int PP_BM = 256, PP_BN = 256, PP_BK = 64, PP_NW = 8;
constexpr int PP_SZ = PP_BM * PP_BK;                      // elements per operand per stage
constexpr int PP_PITCH_C = PP_BN + 8;                     // bf16 epilogue image: 528-byte rows (8-byte stores conflict-free per 16 lanes)
constexpr size_t PP_LDS = (size_t)PP_BM * PP_PITCH_C * 2; // 135168 B >= 4 * PP_SZ * 2 = 131072 B of stages

template <int ACT>
static __device__ __forceinline__ float pp_act(float x) {
  if (ACT == 1) return fmaxf(x, 0.f);
  if (ACT == 2) return gelu_erf(x);
  if (ACT == 3) return tanhf(x);
  return x;
}

template <bool BKS, bool ACONV, bool FAST = false>
__global__ __launch_bounds__(512, 1) void gemm_pp_kernel(GemmParams p) {
  typedef bf16_t T;
  __shared__ __attribute__((aligned(1024))) char smem[PP_LDS];
  T* const sAbase = reinterpret_cast<T*>(smem);
  T* const sBbase = sAbase + 2 * PP_SZ;

  const int tid = threadIdx.x, lane = tid & 63;
#ifndef CMDA_EMU
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: the group test below guards an s_barrier
#else
  const int wid = tid >> 6;
#endif
  const int grp = wid >> 2, wc = wid & 3;                  // M half (= ping-pong group), N quarter
  const int l31 = lane & 31, h = lane >> 5;

  // tile walk: XCD-contiguous ranges, groups of 4 tile rows inside a range (gemm_glds_kernel's walk)
  const int tiles_n = (p.N + PP_BN - 1) / PP_BN;
  const int ntile = gridDim.x;
  int bt = blockIdx.x;
  {
    const int q = ntile / 8, rr = ntile % 8, xcd = bt % 8, loc = bt / 8;
    bt = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  }
  long mt, nt;
  {
    constexpr int GM = 4;
    const int tiles_m = (int)((p.M + PP_BM - 1) / PP_BM);
    if (tiles_n >= 2 * GM && tiles_m >= GM) {
      const int gsz = GM * tiles_n, gid = bt / gsz, first = gid * GM;
      const int gm = min(tiles_m - first, GM), r = bt - gid * gsz;
      mt = first + r % gm;
      nt = r / gm;
    } else {
      mt = bt / tiles_n;
      nt = bt % tiles_n;
    }
  }
  const long m0 = mt * PP_BM, n0 = nt * PP_BN;
  const int z = blockIdx.z;
  const int batch = z / p.batch2, batch2 = z - batch * p.batch2;
  const int nkt = (p.K + PP_BK - 1) / PP_BK;
  const T* baseA = reinterpret_cast<const T*>(p.A.ptr) + (long)batch * p.A.batch_stride + (long)batch2 * p.A.batch2_stride;
  const T* baseB = reinterpret_cast<const T*>(p.B.ptr) + (long)batch * p.B.batch_stride + (long)batch2 * p.B.batch2_stride;

  DmaSrc<false, PP_BM, ACONV, PP_NW, 1, 64, DmaMode<false, ACONV, FAST>::value> dA;   // (FAST: gemm_kernels.h DmaSrc modes)
  DmaSrc<BKS, PP_BN, false, PP_NW, 1, 64, DmaMode<BKS, false, FAST>::value> dB;
  dA.init(p.A, baseA, wid, lane, m0, 0);
  dB.init(p.B, baseB, wid, lane, n0, 0);
  // piece i of k-tile kt into `stage`: pieces 0-3 = A, 4-7 = B (1 KB each, this wave's share of the stage)
  bool cellA = false;
  auto piece = [&](int stage, int kt, int i) {
    // (every (piece, k-tile) is issued exactly once, k-tiles in order: what DmaSrc's running-pointer modes need; piece 0 is the first A
    // piece of its k-tile: the im2col cell test of mode 2 runs there)
    if (i == 0) cellA = dA.cell_begin(kt, 0);
    if (i < 4) glds16(dA.get(p.A, baseA, i, kt, cellA), reinterpret_cast<char*>(sAbase + stage * PP_SZ) + (wid * 4 + i) * 1024);
    else glds16(dB.get(p.B, baseB, i - 4, kt, false), reinterpret_cast<char*>(sBbase + stage * PP_SZ) + (wid * 4 + (i - 4)) * 1024);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment loaders (32x32x16 operand map: lane l -> row / column l & 31, k = 16 s + 8 (l >> 5) + j)
  auto load_a = [&](const T* sA, int mh, u16x8 (&fa)[2][4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = grp * 128 + (mh * 2 + i) * 32 + l31;
#pragma unroll
      for (int s = 0; s < 4; ++s) fa[i][s] = *reinterpret_cast<const u16x8*>(&sA[row * PP_BK + (((2 * s + h) ^ ((row >> 1) & 7)) << 3)]);
    }
  };
  auto load_b = [&](const T* sB, int nh, u16x8 (&fb)[4]) {
    if constexpr (!BKS) {
      const int row = wc * 64 + nh * 32 + l31;
#pragma unroll
      for (int s = 0; s < 4; ++s) fb[s] = *reinterpret_cast<const u16x8*>(&sB[row * PP_BK + (((2 * s + h) ^ ((row >> 1) & 7)) << 3)]);
    } else {
      // [64 k][256 n] image, 32 slots of 16 bytes per line: transposed reads, 4 k x 16 n per 16-lane group
      const int gi = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
      const int colbase = wc * 64 + nh * 32 + 16 * (gi & 1);
      const int cidx = (colbase >> 3) + (pp >> 1), half = (pp & 1) << 2;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int k0 = 16 * s + 8 * (gi >> 1) + q, k1 = k0 + 4;
        const u16x4 lo = lds_read_tr16(&sB[k0 * PP_BN + ((cidx ^ (k0 & 7)) << 3) + half]);
        const u16x4 hi = lds_read_tr16(&sB[k1 * PP_BN + ((cidx ^ (k1 & 7)) << 3) + half]);
        fb[s] = u16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    }
  };

  // prologue: k-tile 0 into stage 0
#pragma unroll
  for (int i = 0; i < 8; ++i) piece(0, 0, i);
  wait_dma_lds();
  raw_barrier();
  if (grp == 1) raw_barrier();     // G1 runs one barrier behind G0 from here on

  // Measured alternative (profiles/r03_pp_phases.txt): the next phase's fragment reads issued right behind the MFMAs and the DMA
  // pieces between the MFMAs -- a piece then costs the issuing wave ~140 cycles of MFMA issue instead of ~60-100 in the LOAD
  // segment (6060 against 4580 cycles per k-tile).
  u16x8 fa[2][4], fb[2][4];   // both N halves of the B fragments stay in registers: phase 3 re-uses phase 0's (no LDS read at all)
  for (int kt = 0; kt < nkt; ++kt) {
    const int st = kt & 1;
    const T* sA = sAbase + st * PP_SZ;
    const T* sB = sBbase + st * PP_SZ;
    const bool more = kt + 1 < nkt;
    // quadrant order (mh, nh) = (0,0) (0,1) (1,1) (1,0): every fragment is read from LDS once per k-tile (A halves in phases 0 / 2,
    // B halves in phases 0 / 1)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int mh = q >> 1, nh = (q == 1 || q == 2) ? 1 : 0;
      // ---- LOAD segment
      PP_STAMP(0);
      if (q == 0 || q == 2) load_a(sA, mh, fa);
      if (q < 2) load_b(sB, nh, fb[nh]);
      if (more) {   // 2 + 4 + 2 pieces: the read-heavy phase 0 (12 ds_read_b128) carries the fewest
        if (q == 0) { piece(st ^ 1, kt + 1, 0); piece(st ^ 1, kt + 1, 4); }
        if (q == 1) { piece(st ^ 1, kt + 1, 1); piece(st ^ 1, kt + 1, 2); piece(st ^ 1, kt + 1, 5); piece(st ^ 1, kt + 1, 6); }
        if (q == 2) { piece(st ^ 1, kt + 1, 3); piece(st ^ 1, kt + 1, 7); }
      }
      // own pieces of k-tile kt+1 landed + this stage's last reads retired: G1 here (its first read of the next stage is two
      // barriers away, G0's one barrier after G1's wait), G0 only at the end of the COMPUTE segment (one more segment of slack)
      if (q == 3 && grp == 1) wait_dma_lds();
      if (q == 3 && grp == 0) wait_lds();
      sched_fence();
      raw_barrier();
      wait_lds();
      PP_STAMP(1);
      sched_fence();
      // ---- COMPUTE segment: one 64 x 32 quadrant x K = 64
      prio(1);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[mh * 2 + i][nh] = mfma_bf16_32x32x16(fb[nh][s], fa[i][s], acc[mh * 2 + i][nh]);
      prio(0);
      PP_STAMP(2);
      if (q == 3 && grp == 0) wait_dma_lds();
      sched_fence();
      raw_barrier();
      PP_STAMP(3);
    }
  }
  if (grp == 0) raw_barrier();     // pairs with G1's last barrier
  __syncthreads();

  // ---- epilogue.  acc[i][j][r]: m = grp*128 + i*32 + (lane & 31), n = wc*64 + j*32 + (r & 3) + 8*(r >> 2) + 4*(lane >> 5)
  T* sC = reinterpret_cast<T*>(smem);
  const float alpha = p.alpha;
  const int act = p.act;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int nl = wc * 64 + j * 32 + 8 * rg + 4 * h;       // 4 consecutive columns nl .. nl+3
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n0 + nl + e < p.N) bv[e] = p.bias[n0 + nl + e];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ml = grp * 128 + i * 32 + l31;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = alpha * acc[i][j][rg * 4 + e] + bv[e];
        if (act == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = pp_act<1>(v[e]);
        } else if (act == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = pp_act<2>(v[e]);
        } else if (act == 3) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = pp_act<3>(v[e]);
        }
        st4(&sC[ml * PP_PITCH_C + nl], v);
      }
    }
  __syncthreads();
  const long cb = (long)batch * p.c_batch_stride + (long)batch2 * p.c_batch2_stride;
  T* C = reinterpret_cast<T*>(p.C) + cb;
  const bool vec = p.c_vec_ok != 0 && (p.ldc & 7) == 0 && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
  // fused column statistics (cmda_gemm_params_t.colstats): a thread stores the same 8-column chunk of every row it handles, so the
  // sums of the STORED (bf16) values ride along in 16 registers -- the accumulators are dead by now
  const bool stats = p.colstats != nullptr;
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, cq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int it = 0; it < (PP_BM * PP_BN / 8) / 512; ++it) {
    const int idx = it * 512 + tid;
    const int row = idx >> 5, ch = idx & 31;
    const long m = m0 + row, n = n0 + ch * 8;
    if (m >= p.M || n >= p.N) continue;
    const uint4 val = *reinterpret_cast<const uint4*>(&sC[row * PP_PITCH_C + ch * 8]);
    if (stats) {
      T tv[8];
      __builtin_memcpy(tv, &val, 16);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = bf2f(tv[e]);
        cs[e] += f;
        cq[e] += f * f;
      }
    }
    if (vec && n + 8 <= p.N) {
      *reinterpret_cast<uint4*>(&C[m * p.ldc + n]) = val;
    } else {
      T tmp[8];
      __builtin_memcpy(tmp, &val, 16);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (n + e < p.N) C[m * p.ldc + n + e] = tmp[e];
    }
  }
  if (stats) {   // lanes l and l ^ 32 hold the same chunk; the eight waves meet in LDS; 2 x 256 atomics per tile (gemm_kernels.h colstats_flush)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      cs[e] += __shfl_xor(cs[e], 32, 64);
      cq[e] += __shfl_xor(cq[e], 32, 64);
    }
    __syncthreads();   // everyone is done with the image
    float* scr = reinterpret_cast<float*>(smem);
    const int wv = tid >> 6, ln = tid & 63;
    if (ln < 32) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        scr[wv * 2 * PP_BN + ln * 8 + e] = cs[e];
        scr[wv * 2 * PP_BN + PP_BN + ln * 8 + e] = cq[e];
      }
    }
    __syncthreads();
    static_assert(2 * PP_BN == 512, "one thread per (statistic, column)");
    const int col = tid & (PP_BN - 1);
    if (n0 + col < p.N) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += scr[w * 2 * PP_BN + tid];
      const long gidx = (unsigned)m0 / (unsigned)p.colstats_rows;
      const unsigned slot = (unsigned)(m0 >> 6) & (CMDA_BN_SLOTS - 1);
      atomicAdd(p.colstats + (gidx * (CMDA_BN_SLOTS + 1) + slot) * 2 * (long)p.N + (tid < PP_BN ? 0 : p.N) + n0 + col, t);
    }
  }
}

}  // namespace

#ifdef CMDA_PP_TIMING
extern "C" int cmda_debug_pp_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_pp_stamps), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -3;
}
#endif

// cross-unit entry (gemm.hip decides eligibility): returns CMDA_ERR_UNSUPPORTED for operand modes this kernel does not take
int cmda_gemm_pp_(const cmda_gemm_params_t& p, void* stream) {
  const long tiles = (long)((p.M + PP_BM - 1) / PP_BM) * ((p.N + PP_BN - 1) / PP_BN);
  const long zz = (long)p.batch * p.batch2;
  if (tiles > 0x7fffffffL || zz > 65535) return CMDA_ERR_SHAPE;
  const dim3 grid((unsigned)tiles, 1, (unsigned)zz), blk(512);
  const bool bks = p.b_kstrided != 0, ac = p.A.conv == 1;
  if (p.a_kstrided || p.B.conv == 1) return CMDA_ERR_UNSUPPORTED;
  const bool nofast = p.tile_hint > 0 && (p.tile_hint & 2048);
  typedef DmaSrc<false, PP_BM, false, PP_NW, 1, 64, 1> FA;
  typedef DmaSrc<false, PP_BM, true, PP_NW, 1, 64, 2> FAC;
  typedef DmaSrc<false, PP_BN, false, PP_NW, 1, 64, 1> FB;
  typedef DmaSrc<true, PP_BN, false, PP_NW, 1, 64, 1> FBK;
  if (!bks && !ac) {
    if (!nofast && FA::mode_ok(p.A, 1) && FB::mode_ok(p.B, 1)) CMDA_LAUNCH((gemm_pp_kernel<false, false, true>), grid, blk, 0, stream, p);
    else CMDA_LAUNCH((gemm_pp_kernel<false, false, false>), grid, blk, 0, stream, p);
  } else if (!bks && ac) {
    if (!nofast && FAC::mode_ok(p.A, 2) && FB::mode_ok(p.B, 1)) CMDA_LAUNCH((gemm_pp_kernel<false, true, true>), grid, blk, 0, stream, p);
    else CMDA_LAUNCH((gemm_pp_kernel<false, true, false>), grid, blk, 0, stream, p);
  } else if (bks && !ac) {
    if (!nofast && FA::mode_ok(p.A, 1) && FBK::mode_ok(p.B, 1)) CMDA_LAUNCH((gemm_pp_kernel<true, false, true>), grid, blk, 0, stream, p);
    else CMDA_LAUNCH((gemm_pp_kernel<true, false, false>), grid, blk, 0, stream, p);
  }
  else return CMDA_ERR_UNSUPPORTED;
  CMDA_CHECK_LAUNCH();
}
