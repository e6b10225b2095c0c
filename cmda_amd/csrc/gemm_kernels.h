// gemm.hip -- MFMA tile GEMM with implicit-im2col operand views and fused epilogues.
//
// One kernel family serves every dense contraction of the CMDA hot path (SURVEY.md K1,K3,K5,K6,K8,K10,K11,K16):
//   nn.Linear q/kv/proj/fc1/fc2 fwd/dgrad/wgrad   mmseg/models/backbones/mix_transformer.py:31-44,62-66,80-102
//   sr / patch-embed / ASPP / bottleneck convs     mix_transformer.py:73-76,169-173; decode_heads/daformer_head.py:63-79
//   attention QK^T, PV and their gradients (batched, strided heads)   mix_transformer.py:97-101
//   CycleGAN generator convs                       mmseg/models/cyclegan/cyclegan_model.py:339-374
//
// C[m,n] = epilogue( alpha * sum_k A(m,k) * B(n,k) ).  Each operand is a *view* V(r,c) that is either a plain
// row-major matrix or an im2col view of an NHWC tensor (r = (b,oh,ow), c = (kh,kw,ci)); no im2col buffer is ever
// materialised.  An operand is used either "K-contiguous" (r = free index, c = k) or "K-strided" (r = k, c = free
// index): the LDS tile always keeps the view's natural orientation so HBM reads stay 16-byte coalesced, and the
// K-strided bf16 fragments come out of LDS through ds_read_b64_tr_b16.
//
// gfx950 tiling: 256 threads = 4 waves (2x2); each wave owns (16*TM)x(16*TN) of the block tile as TMxTN MFMA
// 16x16 accumulators; bf16 uses v_mfma_f32_16x16x32_bf16 (BK=32), f32 uses v_mfma_f32_16x16x4_f32 (BK=16, exact
// fp32 -- the parity mode).  Global->register prefetch of tile k+1 overlaps the MFMAs of tile k.
// Roofline: MFMA-bound for K,N >= 256; HBM-bound below (stage-1/2 Linear layers, C=64/128).
// (kernel templates: included by gemm.hip, gemm_t0..t3.hip and gemm_reg*.hip -- a dozen translation units so that the
// instantiations compile in parallel; every unit keeps its own copies of the anonymous-namespace symbols)
#pragma once
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "../../include/cmda_hip.h"

namespace {

typedef cmda_view_t GemmView;
typedef cmda_gemm_params_t GemmParams;

static __device__ __forceinline__ int reflect_idx(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

// Returns the element offset of V(r, c) or -1 when the element is structural zero (padding / out of range).
static __device__ __forceinline__ long view_offset(const GemmView& v, long r, long c) {
  if (r >= v.R || c >= v.Cc) return -1;
  if (!v.conv) return r * v.ld + c;
  const int ohw = v.OH * v.OW;
  const int b = (int)(r / ohw);
  const int rem = (int)(r - (long)b * ohw);
  const int oh = rem / v.OW;
  const int ow = rem - oh * v.OW;
  const int cell = (int)(c / v.C);
  const int ci = (int)(c - (long)cell * v.C);
  const int kh = cell / v.KW;
  const int kw = cell - kh * v.KW;
  int ih = oh * v.stride - v.pad + kh * v.dil;
  int iw = ow * v.stride - v.pad + kw * v.dil;
  if (v.in_dil > 1) {  // transposed conv: input is zero-inserted by in_dil
    if (ih < 0 || iw < 0 || (ih % v.in_dil) || (iw % v.in_dil)) return -1;
    ih /= v.in_dil;
    iw /= v.in_dil;
  }
  if (v.reflect) {
    ih = reflect_idx(ih, v.H);
    iw = reflect_idx(iw, v.W);
  } else if (ih < 0 || ih >= v.H || iw < 0 || iw >= v.W) {
    return -1;
  }
  return ((long)(b * v.H + ih) * v.W + iw) * v.C + ci;
}

template <typename T>
static __device__ __forceinline__ uint4 load_chunk(const GemmView& v, const T* base, long r, long c) {
  constexpr int CH = Num<T>::kChunk;
  uint4 out = make_uint4(0u, 0u, 0u, 0u);
  if (r >= v.R || c >= v.Cc) return out;
  if (v.vec_ok && c + CH <= v.Cc) {
    const long off = view_offset(v, r, c);
    if (off >= 0) out = *reinterpret_cast<const uint4*>(base + off);
    return out;
  }
  T tmp[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const long off = view_offset(v, r, c + j);
    tmp[j] = off >= 0 ? base[off] : (T)0;
  }
  __builtin_memcpy(&out, tmp, 16);
  return out;
}

// Per-thread staging state of one operand.  Thread t fills chunks id = t + i*256 of the LDS tile: they all sit in the
// same 16-byte column `cc` and in rows row0 + i*RSTEP, so one base offset + one uniform stride describe all of them
// (plain views), and the K loop only adds a constant instead of re-deriving addresses.  PF register sets hold the
// k-tiles in flight (always indexed by compile-time constants).
template <typename T, int NCH, bool KS, int TILE, int BK, int PF>
struct Stager {
  static constexpr int CH = Num<T>::kChunk;
  static constexpr int COLS = KS ? TILE : BK, CPR = COLS / CH, PITCH = COLS + CH, RSTEP = 256 / CPR;
  long off0;           // plain fast path: element offset of chunk 0 for the next k-tile to load
  long istride;        // plain fast path: offset between consecutive chunks of this thread (RSTEP rows)
  long kstep;          // plain fast path: offset between consecutive k-tiles
  int row0, cc;
  int ca[KS ? 1 : NCH], cbc[KS ? 1 : NCH];  // conv fast path constants (see init)
  int cx;
  uint4 reg[PF][NCH];
  bool fast, cfast;

  __device__ __forceinline__ void init(const GemmView& v, int tid, long t0, int kt0) {
    fast = v.vec_ok && !v.conv;
    cfast = v.vec_ok && v.conv && v.in_dil <= 1 && !v.reflect && v.R < (1L << 31) && v.Cc < (1L << 31) &&
            v.H < 32768 && v.W < 32768;
    row0 = tid / CPR;
    cc = tid - row0 * CPR;
    const long r = KS ? (long)kt0 * BK + row0 : t0 + row0;
    const long c = KS ? t0 + cc * CH : (long)kt0 * BK + cc * CH;
    off0 = r * v.ld + c;
    istride = (long)RSTEP * v.ld;
    kstep = KS ? (long)BK * v.ld : (long)BK;
    cx = 0;
#pragma unroll
    for (int i = 0; i < (KS ? 1 : NCH); ++i) ca[i] = cbc[i] = 0;
    if (cfast) {
      if (!KS) {  // the im2col ROW (b,oh,ow) of a chunk never changes: keep b*H and (oh*s-p, ow*s-p); -1 = out of range
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const long ri = r + (long)i * RSTEP;
          if (ri < v.R) {
            const unsigned ohw = (unsigned)(v.OH * v.OW), ru = (unsigned)ri;
            const unsigned b = ru / ohw, rem = ru - b * ohw, oh = rem / (unsigned)v.OW, ow = rem - oh * (unsigned)v.OW;
            ca[i] = (int)b * v.H;
            cbc[i] = (((int)oh * v.stride - v.pad) << 16) | (((int)ow * v.stride - v.pad) & 0xffff);
          } else {
            ca[i] = -1;
          }
        }
      } else {    // the im2col COLUMN (kh,kw,ci) is the same for all chunks of this thread; -1 = out of range
        if (c + CH <= v.Cc) {
          const unsigned cu = (unsigned)c, cell = cu / (unsigned)v.C, ci = cu - cell * (unsigned)v.C;
          const unsigned kh = cell / (unsigned)v.KW, kw = cell - kh * (unsigned)v.KW;
          ca[0] = (int)ci;
          cbc[0] = (((int)kh * v.dil - v.pad) << 16) | (((int)kw * v.dil - v.pad) & 0xffff);
        } else {
          ca[0] = -1;
        }
      }
    }
  }

  template <int S>
  __device__ __forceinline__ void load(const GemmView& v, const T* base, long t0, int kt) {
    const long rbase = KS ? (long)kt * BK + row0 : t0 + row0;
    const long c = KS ? t0 + cc * CH : (long)kt * BK + cc * CH;
    unsigned kh = 0, kw = 0;
    int ci = 0;
    if (cfast && !KS) {  // (kh,kw,ci) of this k-tile's column: same for all chunks of the thread
      const unsigned cu = (unsigned)c, cell = cu / (unsigned)v.C;
      ci = (int)(cu - cell * (unsigned)v.C);
      kh = cell / (unsigned)v.KW;
      kw = cell - kh * (unsigned)v.KW;
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const long r = rbase + (long)i * RSTEP;
      if (fast) {
        if (r < v.R && c + CH <= v.Cc) reg[S][i] = *reinterpret_cast<const uint4*>(base + off0 + (long)i * istride);
        else if (r < v.R && c < v.Cc) reg[S][i] = load_chunk<T>(v, base, r, c);  // ragged last chunk
        else reg[S][i] = make_uint4(0u, 0u, 0u, 0u);
      } else if (cfast && ca[KS ? 0 : i] >= 0 && (KS ? r < v.R : c + CH <= v.Cc)) {
        int bH, ih, iw, cin;
        if (!KS) {
          bH = ca[i];
          ih = (cbc[i] >> 16) + (int)kh * v.dil;
          iw = (int)(short)(cbc[i] & 0xffff) + (int)kw * v.dil;
          cin = ci;
        } else {
          const unsigned ohw = (unsigned)(v.OH * v.OW), ru = (unsigned)r;
          const unsigned b = ru / ohw, rem = ru - b * ohw, oh = rem / (unsigned)v.OW, ow = rem - oh * (unsigned)v.OW;
          cin = ca[0];
          bH = (int)b * v.H;
          ih = (int)oh * v.stride + (cbc[0] >> 16);
          iw = (int)ow * v.stride + (int)(short)(cbc[0] & 0xffff);
        }
        if (ih >= 0 && ih < v.H && iw >= 0 && iw < v.W)
          reg[S][i] = *reinterpret_cast<const uint4*>(base + ((long)(bH + ih) * v.W + iw) * v.C + cin);
        else
          reg[S][i] = make_uint4(0u, 0u, 0u, 0u);
      } else {
        reg[S][i] = load_chunk<T>(v, base, r, c);
      }
    }
    off0 += kstep;
  }

  template <int S>
  __device__ __forceinline__ void store(T* lds) const {
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      *reinterpret_cast<uint4*>(&lds[(row0 + i * RSTEP) * PITCH + cc * CH]) = reg[S][i];
  }
};

// ---------------------------------------------------------------------------------------------------------------
// Fused epilogue, shared by both kernels: C = act(alpha*acc + bias) * rowscale + res + beta*C, read from the fp32 tile
// staged in LDS.  A thread keeps ONE column quad for all its rows, so bias / tail / alignment decisions are made once;
// the row loop is specialised on the activation and has nothing but uniform scalar branches around its loads (the first,
// generic version of this loop spent ~7 us per 128x128 tile in branchy per-element code -- more than the k-loop of the
// short-K GEMMs).
// column of the atomic (split-K / weight-gradient) store: plain, or -- conv weight gradients -- the GEMM column (kh,kw,ci) moved
// to the parameter's own [Co][Ci][KH][KW] layout, so that the gradient needs no staging buffer and no permute-accumulate pass
static __device__ __forceinline__ long atomic_col(const GemmParams& p, long n) {
  if (p.c_perm_ci <= 0) return n;
  const long cell = n / p.c_perm_ci;
  return (n - cell * p.c_perm_ci) * p.c_perm_cells + cell;
}

template <int ACT>
static __device__ __forceinline__ float epi_act(float x) {
  if (ACT == 1) return fmaxf(x, 0.f);
  if (ACT == 2) return gelu_erf(x);
  if (ACT == 3) return tanhf(x);
  return x;
}

template <typename T, int ACT, int BM, int BN, int PITCH_C, int NT>
static __device__ __forceinline__ void epilogue_rows(const GemmParams& p, const float* __restrict__ sC, long m0, long n,
                                                     int q4, int r0, long cb, long rb_off, bool full, const float (&bv)[4],
                                                     float (&cs)[4], float (&cq)[4]) {
  constexpr int QPR = BN / 4, RSTEP = NT / QPR, NIT = (BM + RSTEP - 1) / RSTEP;
  const bool has_res = p.res != nullptr, has_beta = p.beta != 0.f, has_rs = p.rowscale != nullptr, f32o = p.out_f32 != 0;
  const bool res32 = p.res_f32 != 0;
  const float alpha = p.alpha, beta = p.beta;
  const int patch_ow = p.c_patch_ow;
  const long patch_kh = patch_ow > 0 ? n / p.c_patch_kwci : 0, patch_rest = patch_ow > 0 ? n - patch_kh * p.c_patch_kwci : 0;
  const bool stats = p.colstats != nullptr;
#pragma unroll 4
  for (int it = 0; it < NIT; ++it) {
    const int row = r0 + it * RSTEP;
    const long m = m0 + row;
    if (row >= BM || m >= p.M) break;
    const float4 t = *reinterpret_cast<const float4*>(&sC[row * PITCH_C + q4]);
    float v[4] = {t.x, t.y, t.z, t.w};
    long ci = cb + m * p.ldc + n;
    if (patch_ow > 0) {  // un-patchify: ((boh*KH + kh) * OW + ow) * KW*Ci + (kw*Ci + ci)
      const long boh = m / patch_ow;
      const long ow = m - boh * patch_ow;
      ci = ((boh * p.c_patch_kh + patch_kh) * patch_ow + ow) * p.c_patch_kwci + patch_rest;
    }
    const long ri = rb_off + m * p.ldres + n;
    float rv[4] = {0.f, 0.f, 0.f, 0.f}, ov[4] = {0.f, 0.f, 0.f, 0.f};
    float rs = 1.f;
    if (has_rs) rs = p.rowscale[(unsigned)m / (unsigned)p.rows_per_scale];   // (32-bit: row indices are < 2^31 on every GEMM path)
    if (full) {
      if (has_res) {
        if (res32) ld4(reinterpret_cast<const float*>(p.res) + ri, rv);
        else ld4(reinterpret_cast<const T*>(p.res) + ri, rv);
      }
      if (has_beta) {
        if (f32o) ld4(reinterpret_cast<const float*>(p.C) + ci, ov);
        else ld4(reinterpret_cast<const T*>(p.C) + ci, ov);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n + e >= p.N) continue;
        if (has_res) rv[e] = res32 ? reinterpret_cast<const float*>(p.res)[ri + e] : ldf(reinterpret_cast<const T*>(p.res) + ri + e);
        if (has_beta) ov[e] = f32o ? reinterpret_cast<const float*>(p.C)[ci + e] : ldf(reinterpret_cast<const T*>(p.C) + ci + e);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = epi_act<ACT>(alpha * v[e] + bv[e]) * rs + rv[e] + beta * ov[e];
    if (ACT == 0 && stats) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        cs[e] += v[e];
        cq[e] += v[e] * v[e];
      }
    }
    if (full) {
      if (f32o) st4(reinterpret_cast<float*>(p.C) + ci, v);
      else st4(reinterpret_cast<T*>(p.C) + ci, v);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n + e >= p.N) continue;
        if (f32o) reinterpret_cast<float*>(p.C)[ci + e] = v[e];
        else stf(reinterpret_cast<T*>(p.C) + ci + e, v[e]);
      }
    }
  }
}

template <typename T, int BM, int BN, int PITCH_C, int NT = 256>
static __device__ __forceinline__ void epilogue_store(const GemmParams& p, const float* __restrict__ sC, long m0, long n0,
                                                      long cb, long rb_off, int tid, float (&cs)[4], float (&cq)[4]) {
  constexpr int QPR = BN / 4;  // quads per tile row
  static_assert(NT >= QPR, "tile shape");
  // (NT % QPR != 0 -- the 320-column row-panel tile: 80 quads per row, 3 rows per pass, 16 threads idle)
  if (tid >= (NT / QPR) * QPR) return;
  const int q4 = (tid % QPR) * 4, r0 = tid / QPR;
  const long n = n0 + q4;
  if (n >= p.N) return;
  const bool full = p.c_vec_ok != 0 && n + 4 <= p.N;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
    if (full) ld4(p.bias + n, bv);
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < p.N) bv[e] = p.bias[n + e];
    }
  }
  if (p.act == 0) epilogue_rows<T, 0, BM, BN, PITCH_C, NT>(p, sC, m0, n, q4, r0, cb, rb_off, full, bv, cs, cq);
  else if (p.act == 1) epilogue_rows<T, 1, BM, BN, PITCH_C, NT>(p, sC, m0, n, q4, r0, cb, rb_off, full, bv, cs, cq);
  else if (p.act == 2) epilogue_rows<T, 2, BM, BN, PITCH_C, NT>(p, sC, m0, n, q4, r0, cb, rb_off, full, bv, cs, cq);
  else epilogue_rows<T, 3, BM, BN, PITCH_C, NT>(p, sC, m0, n, q4, r0, cb, rb_off, full, bv, cs, cq);
}

// Column statistics of the stored tile (cmda_gemm_params_t.colstats: the BatchNorm / InstanceNorm behind this convolution).  Every
// thread arrives with the sums of ITS column quad over the rows it stored (all epilogue passes of the tile); the lanes of a wave that
// share a quad fold with xor-shuffles, the waves meet in LDS (`scratch`: NT / 64 x 2 x BN floats, the tile staging area, free once
// the last pass is stored -- the caller has a barrier behind it), and thread (statistic, column) issues ONE fp32 atomic into one of
// the 32 slots of the row group's workspace: 2 x BN atomics per tile (a first version issued them per wave and pass -- one atomic per
// four stored elements on the 256 x 256 tile: 255 against 159 us for the head's pointwise convolution).
template <int BN, int NT>
static __device__ __forceinline__ void colstats_flush(const GemmParams& p, float* scratch, long m0, long n0, int tid,
                                                      float (&cs)[4], float (&cq)[4]) {
  constexpr int QPR = BN / 4, NWV = NT / 64;
#pragma unroll
  for (int o = 32; o >= QPR; o >>= 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      cs[e] += __shfl_xor(cs[e], o, 64);
      cq[e] += __shfl_xor(cq[e], o, 64);
    }
  }
  const int lane = tid & 63, wv = tid >> 6;
  const int q4 = (tid % QPR) * 4;
  if (lane < QPR || QPR >= 64) {
    float* d = scratch + wv * 2 * BN;
    st4(d + q4, cs);
    st4(d + BN + q4, cq);
  }
  __syncthreads();
  // (QPR > 64, the 256-wide tile on four waves, would leave quads unowned per wave -- not instantiated: NT >= QPR and the xor fold
  // cover QPR <= 64; the static_assert below keeps it that way)
  static_assert(QPR <= 64, "one wave owns every column quad");
  for (int i = tid; i < 2 * BN; i += NT) {
    const int col = i < BN ? i : i - BN;
    if (n0 + col >= p.N) continue;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) t += scratch[w * 2 * BN + i];
    const long grp = (unsigned)m0 / (unsigned)p.colstats_rows;
    const unsigned slot = (unsigned)(m0 >> 6) & (CMDA_BN_SLOTS - 1);
    atomicAdd(p.colstats + (grp * (CMDA_BN_SLOTS + 1) + slot) * 2 * (long)p.N + (i < BN ? 0 : p.N) + n0 + col, t);
  }
}

template <typename T, int TM, int TN, bool AKS, bool BKS>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
  constexpr int CH = Num<T>::kChunk;
  constexpr int BM = 32 * TM, BN = 32 * TN, BK = 8 * CH;  // BK = 64 (bf16) / 32 (f32)
  constexpr int ROWS_A = AKS ? BK : BM, COLS_A = AKS ? BM : BK, PITCH_A = COLS_A + CH;
  constexpr int ROWS_B = BKS ? BK : BN, COLS_B = BKS ? BN : BK, PITCH_B = COLS_B + CH;
  constexpr int NCH_A = ROWS_A * (COLS_A / CH) / 256, NCH_B = ROWS_B * (COLS_B / CH) / 256;
  static_assert(NCH_A >= 1 && NCH_B >= 1, "tile too small for 256 threads");
  constexpr int SZ_A = ROWS_A * PITCH_A, SZ_B = ROWS_B * PITCH_B;     // elements per stage
  constexpr int PITCH_C = BN + 4;                                       // fp32 epilogue tile
  constexpr size_t STAGE_BYTES = (size_t)2 * (SZ_A + SZ_B) * sizeof(T);
  constexpr size_t EPI_BYTES = (size_t)BM * PITCH_C * sizeof(float);
  constexpr size_t LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;

  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];  // ONE LDS object: 2 stages of A|B, reused by the epilogue
  T* const sAbase = reinterpret_cast<T*>(smem);
  T* const sBbase = sAbase + 2 * SZ_A;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int g = lane >> 4, l15 = lane & 15;

  // Tile order: blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2), so give every XCD a contiguous
  // run of tiles, and inside a run walk the n-tiles of one m-panel first: the A panel (activations, the big operand) is
  // then re-read from that XCD's L2 instead of HBM.  Pure speed heuristic, any placement is correct.
  const int tiles_n = (p.N + BN - 1) / BN;
  const int ntile = gridDim.x;
  int bt = blockIdx.x;
  {
    const int q = ntile / 8, rr = ntile % 8, xcd = bt % 8, loc = bt / 8;
    bt = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  }
  const long m0 = (long)(bt / tiles_n) * BM;
  const long n0 = (long)(bt % tiles_n) * BN;
  const int z = blockIdx.z;
  const int bz = z / p.splits;
  const int split = z - bz * p.splits;
  const int batch = bz / p.batch2;
  const int batch2 = bz - batch * p.batch2;
  const int nkt = (p.K + BK - 1) / BK;
  const int kt_per = (nkt + p.splits - 1) / p.splits;
  const int kt0 = split * kt_per;
  const int kt1 = min(nkt, kt0 + kt_per);

  const T* baseA = reinterpret_cast<const T*>(p.A.ptr) + (long)batch * p.A.batch_stride + (long)batch2 * p.A.batch2_stride;
  const T* baseB = reinterpret_cast<const T*>(p.B.ptr) + (long)batch * p.B.batch_stride + (long)batch2 * p.B.batch2_stride;

  constexpr int PF = (TM * TN <= 8) ? 2 : 1;  // k-tiles in flight in registers (the 128x128 tile has no room for 2)
  Stager<T, NCH_A, AKS, BM, BK, PF> stA;
  Stager<T, NCH_B, BKS, BN, BK, PF> stB;
  stA.init(p.A, tid, m0, kt0);
  stB.init(p.B, tid, n0, kt0);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Software pipeline, two k-tiles ahead: while tile kt is multiplied out of LDS stage kt&1, tile kt+1 sits in one
  // register set (issued an iteration ago) and tile kt+2 is being fetched into the other; ONE barrier per k-tile.
  auto compute = [&](const T* sA, const T* sB) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      u16x8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mr = wm * 16 * TM + i * 16;
        if constexpr (!AKS) {
          fa[i] = *reinterpret_cast<const u16x8*>(&sA[(mr + l15) * PITCH_A + kk * 32 + 8 * g]);
        } else {
          const int q = l15 >> 2, pp = l15 & 3;
          const u16x4 lo = lds_read_tr16(reinterpret_cast<const bf16_t*>(&sA[(kk * 32 + 8 * g + q) * PITCH_A + mr + 4 * pp]));
          const u16x4 hi = lds_read_tr16(reinterpret_cast<const bf16_t*>(&sA[(kk * 32 + 8 * g + 4 + q) * PITCH_A + mr + 4 * pp]));
          fa[i] = u16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int nr = wn * 16 * TN + j * 16;
        if constexpr (!BKS) {
          fb[j] = *reinterpret_cast<const u16x8*>(&sB[(nr + l15) * PITCH_B + kk * 32 + 8 * g]);
        } else {
          const int q = l15 >> 2, pp = l15 & 3;
          const u16x4 lo = lds_read_tr16(reinterpret_cast<const bf16_t*>(&sB[(kk * 32 + 8 * g + q) * PITCH_B + nr + 4 * pp]));
          const u16x4 hi = lds_read_tr16(reinterpret_cast<const bf16_t*>(&sB[(kk * 32 + 8 * g + 4 + q) * PITCH_B + nr + 4 * pp]));
          fb[j] = u16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma_bf16_16x16x32(fa[i], fb[j], acc[i][j]);
    }
  } else {
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      float fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mr = wm * 16 * TM + i * 16;
        fa[i] = AKS ? sA[(ks * 4 + g) * PITCH_A + mr + l15] : sA[(mr + l15) * PITCH_A + ks * 4 + g];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int nr = wn * 16 * TN + j * 16;
        fb[j] = BKS ? sB[(ks * 4 + g) * PITCH_B + nr + l15] : sB[(nr + l15) * PITCH_B + ks * 4 + g];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma_f32_16x16x4(fa[i], fb[j], acc[i][j]);
    }
  }
  };
  if constexpr (PF == 2) {
    if (kt0 < kt1) {
      stA.template load<0>(p.A, baseA, m0, kt0);
      stB.template load<0>(p.B, baseB, n0, kt0);
      stA.template store<0>(sAbase);
      stB.template store<0>(sBbase);
      if (kt0 + 1 < kt1) {
        stA.template load<1>(p.A, baseA, m0, kt0 + 1);
        stB.template load<1>(p.B, baseB, n0, kt0 + 1);
      }
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; kt += 2) {
      if (kt + 2 < kt1) {
        stA.template load<0>(p.A, baseA, m0, kt + 2);
        stB.template load<0>(p.B, baseB, n0, kt + 2);
      }
      compute(sAbase, sBbase);
      if (kt + 1 < kt1) {  // stage 1 was last read one iteration ago, behind that iteration's barrier
        stA.template store<1>(sAbase + SZ_A);
        stB.template store<1>(sBbase + SZ_B);
      }
      __syncthreads();
      if (kt + 1 >= kt1) break;
      if (kt + 3 < kt1) {
        stA.template load<1>(p.A, baseA, m0, kt + 3);
        stB.template load<1>(p.B, baseB, n0, kt + 3);
      }
      compute(sAbase + SZ_A, sBbase + SZ_B);
      if (kt + 2 < kt1) {
        stA.template store<0>(sAbase);
        stB.template store<0>(sBbase);
      }
      __syncthreads();
    }
  } else {
    if (kt0 < kt1) {
      stA.template load<0>(p.A, baseA, m0, kt0);
      stB.template load<0>(p.B, baseB, n0, kt0);
      stA.template store<0>(sAbase);
      stB.template store<0>(sBbase);
    }
    __syncthreads();
    int cur = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
      if (kt + 1 < kt1) {  // next tile's HBM/L2 reads fly behind this tile's MFMAs
        stA.template load<0>(p.A, baseA, m0, kt + 1);
        stB.template load<0>(p.B, baseB, n0, kt + 1);
      }
      compute(sAbase + cur * SZ_A, sBbase + cur * SZ_B);
      if (kt + 1 < kt1) {
        stA.template store<0>(sAbase + (cur ^ 1) * SZ_A);
        stB.template store<0>(sBbase + (cur ^ 1) * SZ_B);
      }
      __syncthreads();
      cur ^= 1;
    }
  }

  if (kt0 >= kt1 && p.splits > 1) return;  // empty split contributes nothing
  const long cb = (long)batch * p.c_batch_stride + (long)batch2 * p.c_batch2_stride;
  const long rb_off = (long)batch * p.res_batch_stride + (long)batch2 * p.res_batch2_stride;

  if (p.atomic) {  // split-K / gradient accumulation: fp32 atomics straight from the accumulators
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const long n = n0 + wn * 16 * TN + j * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long m = m0 + wm * 16 * TM + i * 16 + 4 * g + r;
          if (m < p.M && n < p.N) atomicAdd(reinterpret_cast<float*>(p.C) + cb + m * p.ldc + atomic_col(p, n), p.alpha * acc[i][j][r]);
        }
      }
    return;
  }

  // ---- epilogue through LDS: accumulators (C/D map col = lane&15, row = 4*(lane>>4)+r) -> fp32 tile -> every thread
  //      finishes 4 adjacent columns of one row with vector loads/stores (bias, act, drop-path scale, residual, beta) ----
  float* sC = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        sC[(wm * 16 * TM + i * 16 + 4 * g + r) * PITCH_C + wn * 16 * TN + j * 16 + l15] = acc[i][j][r];
  __syncthreads();
  float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
  epilogue_store<T, BM, BN, PITCH_C>(p, sC, m0, n0, cb, rb_off, tid, cs, cq);
  if (p.colstats) {
    __syncthreads();
    colstats_flush<BN, 256>(p, sC, m0, n0, tid, cs, cq);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA variant (bf16): tiles go HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR staging, no ds_write), 2-4 LDS
// stages (GldsStages), ONE barrier per k-tile that leaves the younger stages' loads in flight.  Works for every operand mode whose 16-byte chunks are aligned (view.vec_ok):
//   K-contiguous tile  [rows][64 k]   : 128-byte lines, fragments by ds_read_b128
//   K-strided   tile   [64 k][cols]   : lines of `cols` bf16, fragments by ds_read_b64_tr_b16
// LDS lines are unpadded (a wave's DMA writes 1 KiB contiguously); the 16-byte slot of logical chunk c of line r is
// c ^ (r & 7), applied to the SOURCE address when filling and to the LDS address when reading.  Anything that must read
// as zero (conv padding, rows/columns past the matrix, the K tail) is fetched from a 16-byte zero block, so no lane
// ever needs a predicated LDS write.
__device__ __attribute__((aligned(16))) unsigned g_zero16[4] = {0u, 0u, 0u, 0u};

// -DCMDA_GEMM_TIMING (tuning builds only): thread 0 of the first 256 blocks stamps the 100 MHz wall clock at the phases
// of the LDS-DMA kernel; tools/gemm_phase.py reads the stamps back through cmda_debug_gemm_stamps().
#ifdef CMDA_GEMM_TIMING
__device__ unsigned long long g_stamps[256 * 8];
#define CMDA_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 256 && blockIdx.z == 0) g_stamps[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define CMDA_STAMP(i) do { } while (0)
#endif

// SWZ = 1 (K-contiguous tiles only): slot = chunk ^ ((line >> 1) & 7) -- conflict-free for ds_read_b128 fragments of 32 consecutive
// lines (the 32x32x16 operand map of gemm_pp.hip: the 16-lane service groups of ds_read_b128 then touch 16 distinct 16-byte positions of
// the 256-byte bank row; with (line & 7) lines r and r + 8 share one)
// BKT: depth of a k-tile (64; 32 for K-STRIDED tiles only -- gemm_wg.hip's four half-depth stages: a K-strided stage is BKT whole lines,
// a K-contiguous one would halve its lines to 64 bytes)
// MODE (compile time, chosen by the host launcher from the operand views -- the state of the other modes is dead code and costs no
// registers: the 8-wave kernels sit at 256 VGPRs per wave):
//   0  general: src() derives every address with its range tests (ragged K, patch views, K-strided im2col, anything)
//   1  FAST, plain operands (no im2col / patch view) whose K extent is a multiple of the k-tile: the zero block is never needed past
//      the first decision -- the source of instruction j is a running pointer plus a constant per k-tile (0 for lanes parked on the
//      zero block).  src() costs ~25 instructions and two branches per DMA instruction per k-tile.
//   2  CELL-FAST, K-contiguous im2col operands whose channel count is a multiple of the k-tile (the head's 3 x 3 bottleneck over 1024
//      channels, the generator's 256-channel convolutions): a k-tile never straddles a (kh, kw) cell, so only the first k-tile of a cell
//      needs the tap arithmetic and the border test -- the other C / 64 - 1 advance the running pointer by 64 channels.
//   3  ROW-FAST, K-strided im2col operands of a stride-1 "same" convolution whose output rows are whole k-tiles (the weight gradient
//      of the head's 3 x 3 bottleneck: 128-pixel rows, 64- or 32-pixel k-tiles): a k-tile is BK consecutive pixels of ONE output row,
//      so (row, segment) are wave-uniform scalars, the pixel index -- and with H == OH, W == OW the source address -- advances by a
//      constant, and the border test is two compares of uniform coordinates against per-lane tap offsets.
template <bool KS, int TILE, bool CONV, int NW = 4, int SWZ = 0, int BKT = 64, int MODE_ = 0>
struct DmaSrc {
  static constexpr int MODE = MODE_;
  static_assert(MODE == 0 || (MODE == 1 && !CONV) || (MODE == 2 && CONV && !KS) || (MODE == 3 && CONV && KS), "DMA source mode");
  static constexpr int BK = BKT;
  static_assert(BKT == 64 || KS, "half-depth k-tiles: K-strided operands only");
  static constexpr int J = TILE * BKT / (512 * NW);                 // DMA instructions per wave per stage
  static constexpr int CPL = KS ? TILE / 8 : 8;                     // 16-byte chunks per LDS line
  static constexpr int LPI = 64 / CPL;                              // lines per DMA instruction
  const bf16_t* ptr[J];   // plain: address of (line, chunk) for k-tile 0, or nullptr when the fixed index is out of range
  int ca[J], cbc[J];      // conv: fixed-index constants
  int fixed[J];           // the fixed coordinate (K-contig: r, K-strided: c), -1 when out of range
  int line[J];            // LDS line of this lane for instruction j
  int chunk[J];           // logical chunk of this lane for instruction j
  // conv views: the moving coordinate of the NEXT k-tile, advanced incrementally by src() (one k-tile = +64 along K; two
  // 32-bit divisions per DMA instruction per k-tile cost about as many VALU cycles as the tile's MFMAs)
  //   K-contiguous (fwd / dgrad):  (ci, kh, kw) of the chunk's first channel      K-strided (wgrad):  (b, oh, ow) of the row
  int s0[J], s1[J], s2[J];
  static constexpr bool conv = CONV;
  const char* cur[J];     // modes 1 / 2: running source of instruction j
  int stepb[J];           // modes 1 / 2: bytes per k-tile (0: parked on the zero block)
  int cellk, cellpos;     // mode 2: k-tiles per (kh, kw) cell; position of the NEXT k-tile inside its cell (wave-uniform)
                          // mode 1, K-contiguous PATCH view: k-tiles per kh segment of KW*C contiguous elements, position inside it
  int jumpb;              // mode 1, K-contiguous patch view: extra bytes when the k-tiles cross into the next kh segment (next input row)
  int dh[J], iw0[J];      // mode 3: tap row offset (kh*dil - pad; out-of-range columns: a value no row can satisfy) and ln + (kw*dil - pad)
  int u_oh, u_seg;        // mode 3: output row / BK-pixel segment of the NEXT k-tile (wave-uniform); c_oh / c_ow0: of the current one
  int c_oh, c_ow0, nseg;

  __device__ __forceinline__ void init(const GemmView& v, const bf16_t* base, int wid, int lane, long t0, int kt0) {
    cellk = MODE == 2 ? v.C / BK : 1;
    cellpos = MODE == 2 ? (int)(((long)kt0 * BK % v.C) / BK) : 0;
    jumpb = 0;
    u_oh = u_seg = c_oh = c_ow0 = 0;
    nseg = 1;
    if (MODE == 3) {
      nseg = v.OW / BK;
      const long row = (long)kt0 / nseg;          // (b*OH + oh) of k-tile kt0
      u_seg = (int)((long)kt0 - row * nseg);
      u_oh = (int)(row % v.OH);
    }
    if (MODE == 1 && !KS && v.conv == 2) {   // patch view, K-contiguous: kh segments of KW*C elements, one input row apart
      const int seg = v.KW * v.C;
      cellk = seg / BK;
      cellpos = (int)(((long)kt0 * BK % seg) / BK);
      jumpb = (int)(((long)v.W * v.C - seg) * (long)sizeof(bf16_t));
    } else if (MODE == 1) {
      cellk = 0x7fffffff;   // plain operand: never crosses a segment
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int ln = (wid * J + j) * LPI + lane / CPL;              // K-contig: tile row; K-strided: k row
      const int slot = lane % CPL;
      line[j] = ln;
      chunk[j] = slot ^ ((SWZ && !KS) ? ((ln >> 1) & 7) : (ln & 7));
      ca[j] = cbc[j] = 0;
      s0[j] = s1[j] = s2[j] = 0;
      ptr[j] = nullptr;
      if (!KS) {
        const long r = t0 + ln;
        fixed[j] = r < v.R ? (int)r : -1;
        if (fixed[j] >= 0) {
          if (!conv) {
            if (v.conv == 2) {  // patch view: row (q = b*OH + oh, ow) starts at q * (s*W*C) + ow * (s*C)
              const unsigned ru = (unsigned)r, q = ru / (unsigned)v.OW, ow = ru - q * (unsigned)v.OW;
              ptr[j] = base + (long)q * v.stride * v.W * v.C + (long)ow * v.stride * v.C + chunk[j] * 8;
            } else {
              ptr[j] = base + r * v.ld + chunk[j] * 8;
            }
          } else {
            const unsigned ohw = (unsigned)(v.OH * v.OW), ru = (unsigned)r;
            const unsigned b = ru / ohw, rem = ru - b * ohw, oh = rem / (unsigned)v.OW, ow = rem - oh * (unsigned)v.OW;
            ca[j] = (int)b * v.H;
            cbc[j] = (((int)oh * v.stride - v.pad) << 16) | (((int)ow * v.stride - v.pad) & 0xffff);
            const unsigned cu = (unsigned)((long)kt0 * BK + chunk[j] * 8), cell = cu / (unsigned)v.C;
            s0[j] = (int)(cu - cell * (unsigned)v.C);
            s1[j] = (int)(cell / (unsigned)v.KW);
            s2[j] = (int)(cell - (unsigned)s1[j] * (unsigned)v.KW);
          }
        }
      } else {
        const long c = t0 + chunk[j] * 8;
        fixed[j] = c + 8 <= v.Cc ? (int)c : -1;
        if (fixed[j] >= 0) {
          if (!conv) {
            if (v.conv == 2) {  // patch view: column (kh, jj) sits at kh * (W*C) + jj; the pixel row advances with the k-tiles
              const unsigned seg = (unsigned)(v.KW * v.C), cu = (unsigned)c, kh = cu / seg;
              ptr[j] = base + (long)kh * v.W * v.C + (cu - kh * seg);
              const unsigned ru = (unsigned)((long)kt0 * BK + ln);
              s1[j] = (int)(ru / (unsigned)v.OW);
              s2[j] = (int)(ru - (unsigned)s1[j] * (unsigned)v.OW);
            } else {
              ptr[j] = base + (long)ln * v.ld + c;
            }
          } else {
            const unsigned cu = (unsigned)c, cell = cu / (unsigned)v.C, ci = cu - cell * (unsigned)v.C;
            const unsigned kh = cell / (unsigned)v.KW, kw = cell - kh * (unsigned)v.KW;
            ca[j] = (int)ci;
            cbc[j] = (((int)kh * v.dil - v.pad) << 16) | (((int)kw * v.dil - v.pad) & 0xffff);
            const unsigned ohw = (unsigned)(v.OH * v.OW), ru = (unsigned)((long)kt0 * BK + ln);
            const unsigned b = ru / ohw, rem = ru - b * ohw;
            s0[j] = (int)b;
            s1[j] = (int)(rem / (unsigned)v.OW);
            s2[j] = (int)(rem - (unsigned)s1[j] * (unsigned)v.OW);
          }
        }
      }
      if (MODE == 3) {   // pixel r0 = kt0*BK + ln under tap (dh, dw): element (r0 + dh*W + dw)*C + ci -- linear in r0 (H == OH, W == OW)
        if (fixed[j] >= 0) {
          dh[j] = cbc[j] >> 16;
          const int dw = (int)(short)(cbc[j] & 0xffff);
          iw0[j] = ln + dw;
          cur[j] = reinterpret_cast<const char*>(base + ((long)kt0 * BK + ln + (long)dh[j] * v.W + dw) * v.C + ca[j]);
        } else {
          dh[j] = 1 << 24;
          iw0[j] = 0;
          cur[j] = reinterpret_cast<const char*>(g_zero16);
        }
        stepb[j] = (int)((long)BK * v.C * (long)sizeof(bf16_t));
      } else if (MODE == 1) {   // running source: k-tile kt0 of this lane's (line, chunk), or the zero block (step 0)
        if (fixed[j] >= 0 && ptr[j] != nullptr) {
          if (v.conv == 2 && !KS) {          // patch view, K-contiguous: segment kh0 of k-tile kt0, offset inside it
            const long c0 = (long)kt0 * BK, seg = (long)v.KW * v.C, kh0 = c0 / seg;
            stepb[j] = BK * (int)sizeof(bf16_t);
            cur[j] = reinterpret_cast<const char*>(ptr[j] + kh0 * v.W * v.C + (c0 - kh0 * seg));
          } else if (v.conv == 2) {          // patch view, K-strided, OW divides the k-tile: this lane's ow never changes and its
                                             // pixel row q advances BK / OW rows per k-tile -- a constant step
            stepb[j] = (int)((long)(BK / v.OW) * v.stride * v.W * v.C * (long)sizeof(bf16_t));
            cur[j] = reinterpret_cast<const char*>(ptr[j] + (long)s1[j] * v.stride * v.W * v.C + (long)s2[j] * v.stride * v.C);
          } else {
            stepb[j] = (int)((KS ? (long)BK * v.ld : (long)BK) * (long)sizeof(bf16_t));
            cur[j] = reinterpret_cast<const char*>(ptr[j]) + (long)kt0 * stepb[j];
          }
        } else {
          stepb[j] = 0;
          cur[j] = reinterpret_cast<const char*>(g_zero16);
        }
      } else {
        stepb[j] = 0;
        cur[j] = nullptr;
      }
    }
  }

  // cell-fast path, first k-tile issued by this block or first k-tile of a (kh, kw) cell: full address (channel offset of the
  // k-tile inside the cell included), later k-tiles of the cell: cell_next
  __device__ __forceinline__ const void* cell_first(const GemmView& v, const bf16_t* base, int j, int kt) {
    const char* zero = reinterpret_cast<const char*>(g_zero16);
    cur[j] = zero;
    stepb[j] = 0;
    if (fixed[j] < 0) return zero;
    const unsigned c0 = (unsigned)(kt * BK), cell = c0 / (unsigned)v.C, ci0 = c0 - cell * (unsigned)v.C;   // wave-uniform
    const int kh = (int)(cell / (unsigned)v.KW), kw = (int)(cell - (unsigned)kh * (unsigned)v.KW);
    int ih = (cbc[j] >> 16) + kh * v.dil, iw = (int)(short)(cbc[j] & 0xffff) + kw * v.dil;
    if (v.in_dil > 1) {
      if (ih < 0 || iw < 0) return zero;
      if (v.in_dil == 2) {
        if ((ih | iw) & 1) return zero;
        ih >>= 1;
        iw >>= 1;
      } else {
        if ((ih % v.in_dil) || (iw % v.in_dil)) return zero;
        ih /= v.in_dil;
        iw /= v.in_dil;
      }
    }
    if (v.reflect) {
      ih = reflect_idx(ih, v.H);
      iw = reflect_idx(iw, v.W);
    } else if (ih < 0 || ih >= v.H || iw < 0 || iw >= v.W) {
      return zero;
    }
    cur[j] = reinterpret_cast<const char*>(base + ((long)(ca[j] + ih) * v.W + iw) * v.C + (int)ci0 + chunk[j] * 8);
    stepb[j] = BK * (int)sizeof(bf16_t);
    return cur[j];
  }
  __device__ __forceinline__ const void* cell_next(int j) {
    cur[j] += stepb[j];
    return cur[j];
  }

  // mode 1: source of instruction j for the NEXT k-tile in issue order (kt0, kt0 + 1, ...)
  __device__ __forceinline__ const void* next(int j) {
    const char* s = cur[j];
    cur[j] = s + stepb[j];
    return s;
  }

  // call ONCE per k-tile (before the get() calls of that k-tile).  Mode 2: is it the first k-tile of its cell / of this block?
  // Mode 1 on a K-contiguous patch view: did the PREVIOUS k-tile end a kh segment (the running pointers then jump to the next row)?
  __device__ __forceinline__ bool cell_begin(int kt, int kt0, const GemmView* v = nullptr) {
    if (MODE == 3) {   // uniform coordinates of this k-tile; advance to the next
      c_oh = u_oh;
      c_ow0 = u_seg * BK;
      if (++u_seg == nseg) {
        u_seg = 0;
        if (++u_oh == v->OH) u_oh = 0;
      }
      return false;
    }
    if (MODE == 1) {
      if (KS) return false;
      const bool cross = cellpos == cellk;   // (plain operands: cellk = INT_MAX, never)
      cellpos = cross ? 1 : cellpos + 1;
      return cross;
    }
    if (MODE != 2) return false;
    const bool first = cellpos == 0 || kt == kt0;
    cellpos = cellpos + 1 == cellk ? 0 : cellpos + 1;
    return first;
  }
  // every mode: source of instruction j of k-tile kt; every (j, k-tile) exactly once, k-tiles in order
  __device__ __forceinline__ const void* get(const GemmView& v, const bf16_t* base, int j, int kt, bool first) {
    if (MODE == 1) {
      if (!KS && first) cur[j] += stepb[j] ? jumpb : 0;
      return next(j);
    }
    if (MODE == 2) return first ? cell_first(v, base, j, kt) : cell_next(j);
    if (MODE == 3) {
      const char* s = cur[j];
      cur[j] = s + stepb[j];
      const bool ok = (unsigned)(c_oh + dh[j]) < (unsigned)v.H && (unsigned)(c_ow0 + iw0[j]) < (unsigned)v.W;
      return ok ? s : reinterpret_cast<const char*>(g_zero16);
    }
    return src(v, base, j, kt);
  }
  // HOST: may this view run in mode 1 (plain) / 2 (K-contiguous im2col)?
  static bool mode_ok(const GemmView& v, int mode) {
    if (mode == 1) {
      if (((KS ? v.R : v.Cc) % BKT) != 0 || v.conv == 1) return false;
      if (v.conv == 0) return (KS ? (long)BKT * v.ld * 2 : (long)BKT * 2) < (1L << 31);
      // patch views (kernel == stride convolutions: the spatial-reduction convs): K-contiguous always (KW*C % 64 == 0 is a condition of
      // the LDS-DMA path), K-strided when the output row length divides the k-tile
      if (!KS) return ((long)v.KW * v.C) % BKT == 0 && (long)v.W * v.C * 2 < (1L << 31);
      return v.OW > 0 && (BKT % v.OW) == 0 && (long)(BKT / v.OW) * v.stride * v.W * v.C * 2 < (1L << 31);
    }
    if (mode == 2) return v.conv == 1 && !KS && v.vec_ok && (v.C % BKT) == 0;
    if (mode == 3) return v.conv == 1 && KS && v.vec_ok && v.stride == 1 && v.in_dil <= 1 && !v.reflect && v.H == v.OH && v.W == v.OW &&
                          (v.OW % BKT) == 0 && (v.R % BKT) == 0 && (long)BKT * v.C * 2 < (1L << 31);
    return true;
  }

  // source address of this lane's 16 bytes for instruction j of k-tile kt.  Conv views: must be called once per (j, kt)
  // with kt = kt0, kt0 + 1, ... (the pipeline issues k-tiles in order) -- the call advances the moving coordinate.
  __device__ __forceinline__ const void* src(const GemmView& v, const bf16_t* base, int j, int kt) {
    const void* zero = reinterpret_cast<const void*>(g_zero16);
    if (fixed[j] < 0) return zero;
    if (!KS) {
      const long c = (long)kt * BK + chunk[j] * 8;
      if (!conv) {
        if (c + 8 > v.Cc) return zero;
        if (v.conv == 2) {  // k-tile kt lies inside one (kh) segment of KW*C contiguous elements (KW*C % 64 == 0)
          const unsigned seg = (unsigned)(v.KW * v.C), c0 = (unsigned)(kt * BK), kh = c0 / seg;
          return ptr[j] + (long)kh * v.W * v.C + (c0 - kh * seg);
        }
        return ptr[j] + (long)kt * BK;
      }
      const int ci = s0[j], kh = s1[j], kw = s2[j];
      {  // advance to the next k-tile: +64 channels, carrying into (kw, kh)
        int nci = ci + BK, nkw = kw, nkh = kh;
        while (nci >= v.C) {
          nci -= v.C;
          if (++nkw == v.KW) { nkw = 0; ++nkh; }
        }
        s0[j] = nci; s1[j] = nkh; s2[j] = nkw;
      }
      if (c + 8 > v.Cc) return zero;
      int ih = (cbc[j] >> 16) + kh * v.dil, iw = (int)(short)(cbc[j] & 0xffff) + kw * v.dil;
      if (v.in_dil > 1) {   // zero-inserted input (transposed convolution / the data gradient of a strided one): taps that fall
                            // between the real pixels read zero
        if (ih < 0 || iw < 0) return zero;
        if (v.in_dil == 2) {
          if ((ih | iw) & 1) return zero;
          ih >>= 1;
          iw >>= 1;
        } else {
          if ((ih % v.in_dil) || (iw % v.in_dil)) return zero;
          ih /= v.in_dil;
          iw /= v.in_dil;
        }
      }
      if (v.reflect) {
        ih = reflect_idx(ih, v.H);
        iw = reflect_idx(iw, v.W);
      } else if (ih < 0 || ih >= v.H || iw < 0 || iw >= v.W) {
        return zero;
      }
      return base + ((long)(ca[j] + ih) * v.W + iw) * v.C + ci;
    } else {
      const long r = (long)kt * BK + line[j];
      if (!conv) {
        if (v.conv == 2) {
          const int q = s1[j], ow = s2[j];
          {  // advance to the next k-tile: +64 output pixels, carrying into q = b*OH + oh
            int now = ow + BK, nq = q;
            while (now >= v.OW) { now -= v.OW; ++nq; }
            s1[j] = nq; s2[j] = now;
          }
          return r >= v.R ? zero : static_cast<const void*>(ptr[j] + (long)q * v.stride * v.W * v.C + (long)ow * v.stride * v.C);
        }
        return r >= v.R ? zero : static_cast<const void*>(ptr[j] + (long)kt * BK * v.ld);
      }
      const int b = s0[j], oh = s1[j], ow = s2[j];
      {  // advance to the next k-tile: +64 output pixels, carrying into (oh, b)
        int now = ow + BK, noh = oh, nb = b;
        while (now >= v.OW) {
          now -= v.OW;
          if (++noh == v.OH) { noh = 0; ++nb; }
        }
        s0[j] = nb; s1[j] = noh; s2[j] = now;
      }
      if (r >= v.R) return zero;
      int ih = oh * v.stride + (cbc[j] >> 16), iw = ow * v.stride + (int)(short)(cbc[j] & 0xffff);
      if (v.reflect) {
        ih = reflect_idx(ih, v.H);
        iw = reflect_idx(iw, v.W);
      } else if (ih < 0 || ih >= v.H || iw < 0 || iw >= v.W) {
        return zero;
      }
      return base + ((long)(b * v.H + ih) * v.W + iw) * v.C + ca[j];
    }
  }
};

// LDS stages per tile shape.  Measured (profiles/README.md, round 1 k): 4 stages on the 64x64 tile cut the k-loop of a
// lone block from ~5 to 1.6 us, but 64 KiB of LDS halves the blocks per CU and the whole step got slower (the loop is
// bound by the per-CU L2->LDS rate, ~65 GB/s, which wants MORE resident blocks, not deeper ones) -- so every tile keeps 2.
template <int TM, int TN> struct GldsStages { static constexpr int value = 2; };

// NW = 4 waves (2 x 2): tiles 64x64 / 128x64 / 128x128.  NW = 8 waves (4 x 2, 512 threads, one block per CU): the 256x256
// tile for the large GEMMs -- the k-loop is bound by the per-CU L2->LDS rate, and a 256x256 tile moves half the bytes
// per FLOP of a 128x128 one.  Its epilogue goes through LDS one wave-row (64 rows) at a time.
// NSV = 0: the throughput configuration (GldsStages: 2 stages, most resident blocks -- right when the grid is many waves of
// blocks deep).  NSV = 4: the LATENCY configuration for grids that fit the chip in one wave anyway (the 2 + 2-sample UDA step:
// every Linear of an encoder is 160-320 blocks): with 2 stages every k-tile exposes a full load latency (K = 1280: 8.9 us of an
// 11.4 us kernel, tools/gemm_phase.py), with 4 stages three tiles are in flight and the lone block's k-loop drops ~3x.
template <int TM, int TN, int NW, int NSV>
struct GldsCfg {
  static constexpr int WM = NW / 2, NTHR = 64 * NW;
  static constexpr int BM = 16 * TM * WM, BN = 32 * TN, BK = 64;
  static constexpr int NS = NSV ? NSV : GldsStages<TM, TN>::value;
  static constexpr int SZ_A = BM * BK, SZ_B = BN * BK;                       // elements per stage
  static constexpr int PITCH_C = BN + 4;
  static constexpr int EPI_ROWS = NW == 8 ? 16 * TM : BM;                    // rows staged per epilogue pass
  static constexpr size_t STAGE_BYTES = (size_t)NS * (SZ_A + SZ_B) * sizeof(bf16_t);
  static constexpr size_t EPI_BYTES = (size_t)EPI_ROWS * PITCH_C * sizeof(float);
  static constexpr size_t LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
  static constexpr int MIN_WAVES = (NW == 8 || NSV * (TM * 16 * (NW / 2) + TN * 32) * 128 > 65536) ? 1 : 2;
};

// The kernel body.  (bt_raw, ntile, z) = this workgroup's tile index, the number of tiles of the problem and its (batch, split)
// index: blockIdx.x / gridDim.x / blockIdx.z of a plain launch, or read from the block map of a GROUPED launch (many problems
// of the same template instance in one grid -- the deferred weight gradients of a backward pass, gemm_glds_grouped_kernel).
// FAST: every operand in its fast DMA-source mode (plain -> 1, K-contiguous im2col -> 2; a K-strided im2col operand stays general)
template <bool KS, bool CONV, bool FAST> struct DmaMode { static constexpr int value = !FAST ? 0 : (CONV ? (KS ? 3 : 2) : 1); };

template <int TM, int TN, bool AKS, bool BKS, bool ACONV, bool BCONV, int NW, int NSV, bool FAST = false>
static __device__ __forceinline__ void gemm_glds_body(const GemmParams& p, char* smem, int bt_raw, int ntile_raw, int z_raw, bool xcd_walk) {
  typedef bf16_t T;
  typedef GldsCfg<TM, TN, NW, NSV> Cfg;
  constexpr int WM = Cfg::WM, NTHR = Cfg::NTHR, BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, NS = Cfg::NS;
  constexpr int SZ_A = Cfg::SZ_A, SZ_B = Cfg::SZ_B, PITCH_C = Cfg::PITCH_C, EPI_ROWS = Cfg::EPI_ROWS;
  T* const sAbase = reinterpret_cast<T*>(smem);
  T* const sBbase = sAbase + NS * SZ_A;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, g = lane >> 4, l15 = lane & 15;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int ntile = ntile_raw;
  int bt = bt_raw;
  if (xcd_walk) {
    const int q = ntile / 8, rr = ntile % 8, xcd = bt % 8, loc = bt / 8;
    bt = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  }
  // Inside an XCD's contiguous range the tiles are walked in groups of GM tile rows, column by column: the ~32 (one block per
  // CU) or ~64 blocks an XCD runs at once then cover a GM x (32/GM) patch of the output instead of one row of it, and share
  // GM A panels + 32/GM B panels through that XCD's L2 instead of 1 + 32 (8192^3 on the 256x256 tile: 2 MB of DMA per k-tile
  // per XCD of which 1.06 MB unique row-major, 0.38 MB unique grouped -- the row-major walk ran at the memory-side rate).
  long mt, nt;
  {
    constexpr int GM = NW == 8 ? 4 : 8;
    const int tiles_m = (int)((p.M + BM - 1) / BM);
    if (tiles_n >= 2 * GM && tiles_m >= GM && !(p.tile_hint > 0 && (p.tile_hint & 256))) {
      const int gsz = GM * tiles_n, gid = bt / gsz, first = gid * GM;
      const int gm = min(tiles_m - first, GM), r = bt - gid * gsz;
      mt = first + r % gm;
      nt = r / gm;
    } else {
      mt = bt / tiles_n;
      nt = bt % tiles_n;
    }
  }
  const long m0 = mt * BM;
  const long n0 = nt * BN;
  const int z = z_raw;
  const int bz = z / p.splits;
  const int split = z - bz * p.splits;
  const int batch = bz / p.batch2, batch2 = bz - batch * p.batch2;
  const int nkt = (p.K + BK - 1) / BK;
  const int kt_per = (nkt + p.splits - 1) / p.splits;
  const int kt0 = split * kt_per;
  const int kt1 = min(nkt, kt0 + kt_per);
  const T* baseA = reinterpret_cast<const T*>(p.A.ptr) + (long)batch * p.A.batch_stride + (long)batch2 * p.A.batch2_stride;
  const T* baseB = reinterpret_cast<const T*>(p.B.ptr) + (long)batch * p.B.batch_stride + (long)batch2 * p.B.batch2_stride;

  CMDA_STAMP(0);
  DmaSrc<AKS, BM, ACONV, NW, 0, 64, DmaMode<AKS, ACONV, FAST>::value> dA;
  DmaSrc<BKS, BN, BCONV, NW, 0, 64, DmaMode<BKS, BCONV, FAST>::value> dB;
  dA.init(p.A, baseA, wid, lane, m0, kt0);
  dB.init(p.B, baseB, wid, lane, n0, kt0);
  auto issue = [&](int stage, int kt) {
    char* la = reinterpret_cast<char*>(sAbase + stage * SZ_A) + wid * dA.J * 1024;
    char* lb = reinterpret_cast<char*>(sBbase + stage * SZ_B) + wid * dB.J * 1024;
    const bool fa = dA.cell_begin(kt, kt0, &p.A), fb = dB.cell_begin(kt, kt0, &p.B);
#pragma unroll
    for (int j = 0; j < dA.J; ++j) glds16(dA.get(p.A, baseA, j, kt, fa), la + j * 1024);
#pragma unroll
    for (int j = 0; j < dB.J; ++j) glds16(dB.get(p.B, baseB, j, kt, fb), lb + j * 1024);
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // optional fused bias gradient: blocks of the first n-tile also add up their A tile along k (A is dY^T here)
  const bool do_colsum = AKS && p.colsum != nullptr && n0 == 0;
  float bsum = 0.f;
  constexpr int LPT = DmaSrc<AKS, BM, ACONV, NW>::J + DmaSrc<BKS, BN, BCONV, NW>::J;  // DMA instructions per wave per k-tile
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (kt0 + s < kt1) issue(s, kt0 + s);
  CMDA_STAMP(1);
  int st = 0;
  for (int kt = kt0; kt < kt1; ++kt) {
    // tile kt has landed: in steady state NS-2 younger tiles may stay in flight; in the tail fewer were issued, so drain.
    // Past the barrier every wave is done reading stage (st-1), which the next DMA overwrites.
    if (kt + NS - 2 < kt1) pipe_barrier<(NS - 2) * LPT>();
    else pipe_barrier<0>();
    if (kt == kt0) CMDA_STAMP(2);
    {
      int sn = st + NS - 1;
      if (sn >= NS) sn -= NS;
      if (kt + NS - 1 < kt1) issue(sn, kt + NS - 1);
    }
    const T* sA = sAbase + st * SZ_A;
    const T* sB = sBbase + st * SZ_B;
    if constexpr (AKS) {
      if (do_colsum && tid < BM) {
#pragma unroll 8
        for (int k = 0; k < BK; ++k) bsum += bf2f(sA[k * BM + (((tid >> 3) ^ (k & 7)) << 3) + (tid & 7)]);
      }
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      u16x8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mr = wm * 16 * TM + i * 16;
        if constexpr (!AKS) {
          const int row = mr + l15;
          fa[i] = *reinterpret_cast<const u16x8*>(&sA[row * BK + (((kk * 4 + g) ^ (row & 7)) << 3)]);
        } else {
          const int q = l15 >> 2, pp = l15 & 3;
          const int k0 = kk * 32 + 8 * g + q, k1 = k0 + 4;
          const int cidx = (mr >> 3) + (pp >> 1), half = (pp & 1) << 2;
          const u16x4 lo = lds_read_tr16(&sA[k0 * BM + ((cidx ^ (k0 & 7)) << 3) + half]);
          const u16x4 hi = lds_read_tr16(&sA[k1 * BM + ((cidx ^ (k1 & 7)) << 3) + half]);
          fa[i] = u16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int nr = wn * 16 * TN + j * 16;
        if constexpr (!BKS) {
          const int row = nr + l15;
          fb[j] = *reinterpret_cast<const u16x8*>(&sB[row * BK + (((kk * 4 + g) ^ (row & 7)) << 3)]);
        } else {
          const int q = l15 >> 2, pp = l15 & 3;
          const int k0 = kk * 32 + 8 * g + q, k1 = k0 + 4;
          const int cidx = (nr >> 3) + (pp >> 1), half = (pp & 1) << 2;
          const u16x4 lo = lds_read_tr16(&sB[k0 * BN + ((cidx ^ (k0 & 7)) << 3) + half]);
          const u16x4 hi = lds_read_tr16(&sB[k1 * BN + ((cidx ^ (k1 & 7)) << 3) + half]);
          fb[j] = u16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma_bf16_16x16x32(fa[i], fb[j], acc[i][j]);
    }
    if (++st == NS) st = 0;
  }
  CMDA_STAMP(3);
  __syncthreads();

  if (kt0 >= kt1 && p.splits > 1) return;
  if constexpr (AKS) {
    if (do_colsum && tid < BM && m0 + tid < p.M) atomicAdd(p.colsum + m0 + tid, bsum);
  }
  const long cb = (long)batch * p.c_batch_stride + (long)batch2 * p.c_batch2_stride;
  const long rb_off = (long)batch * p.res_batch_stride + (long)batch2 * p.res_batch2_stride;
  if (p.atomic) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const long n = n0 + wn * 16 * TN + j * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long m = m0 + wm * 16 * TM + i * 16 + 4 * g + r;
          if (m < p.M && n < p.N) atomicAdd(reinterpret_cast<float*>(p.C) + cb + m * p.ldc + atomic_col(p, n), p.alpha * acc[i][j][r]);
        }
      }
    return;
  }
  float* sC = reinterpret_cast<float*>(smem);
  if constexpr (NW == 4) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          sC[(wm * 16 * TM + i * 16 + 4 * g + r) * PITCH_C + wn * 16 * TN + j * 16 + l15] = acc[i][j][r];
    __syncthreads();
    CMDA_STAMP(4);
    float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
    epilogue_store<T, BM, BN, PITCH_C, NTHR>(p, sC, m0, n0, cb, rb_off, tid, cs, cq);
    if (p.colstats) {
      __syncthreads();
      colstats_flush<BN, NTHR>(p, sC, m0, n0, tid, cs, cq);
    }
  } else {
    float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
    for (int wr = 0; wr < WM; ++wr) {  // one wave-row (16*TM rows) per pass: its two waves stage, everyone stores
      if (wm == wr) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              sC[(i * 16 + 4 * g + r) * PITCH_C + wn * 16 * TN + j * 16 + l15] = acc[i][j][r];
      }
      __syncthreads();
      if (m0 + wr * EPI_ROWS < p.M)
        epilogue_store<T, EPI_ROWS, BN, PITCH_C, NTHR>(p, sC, m0 + wr * EPI_ROWS, n0, cb, rb_off, tid, cs, cq);
      __syncthreads();
    }
    if (p.colstats) colstats_flush<BN, NTHR>(p, sC, m0, n0, tid, cs, cq);
    CMDA_STAMP(4);
  }
  CMDA_STAMP(5);
}

template <int TM, int TN, bool AKS, bool BKS, bool ACONV, bool BCONV, int NW = 4, int NSV = 0, bool FAST = false>
__global__ __launch_bounds__(64 * NW, (GldsCfg<TM, TN, NW, NSV>::MIN_WAVES)) void gemm_glds_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(1024))) char smem[GldsCfg<TM, TN, NW, NSV>::LDS_BYTES];
  gemm_glds_body<TM, TN, AKS, BKS, ACONV, BCONV, NW, NSV, FAST>(p, smem, blockIdx.x, gridDim.x, blockIdx.z, true);
}

// GROUPED launch: one grid over MANY problems of this template instance.  `tab` = DEVICE array of parameter blocks (splits already
// resolved), `blk` = DEVICE array with one {problem, block index inside the problem} pair per workgroup (problem < 0: padding,
// the workgroup exits); inside a problem the blocks are numbered z-major (z * tiles + tile).  The planner (gemm_grouped.hip)
// lays the map out so that the workgroups an XCD receives (b, b + 8, b + 16, ...) walk whole (problem, split) units: their
// tiles read the same K slice of both operands through that XCD's L2.  Used for the DEFERRED weight gradients of a backward pass (ops.gemm_deferral):
// ~300 latency-bound launches of 100-200 blocks each per encoder stage become one launch of ~50 k blocks that runs at the
// MFMA / atomic rate.  The XCD-aware tile walk is off (weight-gradient outputs are a few tiles; nothing to share through L2).
template <int TM, int TN, bool AKS, bool BKS, bool ACONV, bool BCONV, int NW = 4, int NSV = 0, bool FAST = false>
__global__ __launch_bounds__(64 * NW, (GldsCfg<TM, TN, NW, NSV>::MIN_WAVES)) void gemm_glds_grouped_kernel(const GemmParams* __restrict__ tab,
                                                                                                            const int* __restrict__ blk) {
  __shared__ __attribute__((aligned(1024))) char smem[GldsCfg<TM, TN, NW, NSV>::LDS_BYTES];
#ifndef CMDA_EMU
  const int prob = __builtin_amdgcn_readfirstlane(blk[2 * blockIdx.x]), loc = __builtin_amdgcn_readfirstlane(blk[2 * blockIdx.x + 1]);
#else
  const int prob = blk[2 * blockIdx.x], loc = blk[2 * blockIdx.x + 1];
#endif
  if (prob < 0) return;   // padding entry of the XCD-interleaved block map
  const GemmParams& p = tab[prob];
  constexpr int BM = GldsCfg<TM, TN, NW, NSV>::BM, BN = GldsCfg<TM, TN, NW, NSV>::BN;
  const int ntile = (int)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const int z = loc / ntile;
  gemm_glds_body<TM, TN, AKS, BKS, ACONV, BCONV, NW, NSV, FAST>(p, smem, loc - z * ntile, ntile, z, false);
}

template <int TM, int TN, int NW, int NSV, bool AKS, bool BKS, bool ACONV, bool BCONV>
int launch_glds_mode(const GemmParams& p, const dim3& grid, void* stream) {
  constexpr int BM = 16 * TM * (NW / 2), BN = 32 * TN;
  const dim3 blk(64 * NW);
  typedef DmaSrc<AKS, BM, ACONV, NW, 0, 64, DmaMode<AKS, ACONV, true>::value> FA;
  typedef DmaSrc<BKS, BN, BCONV, NW, 0, 64, DmaMode<BKS, BCONV, true>::value> FB;
  const bool fast = FA::mode_ok(p.A, FA::MODE) && FB::mode_ok(p.B, FB::MODE) && (FA::MODE != 0 || FB::MODE != 0) &&
                    !(p.tile_hint > 0 && (p.tile_hint & 2048));   // (tile_hint bit 11: force the general address path, tuning A/B)
  if (fast) CMDA_LAUNCH((gemm_glds_kernel<TM, TN, AKS, BKS, ACONV, BCONV, NW, NSV, true>), grid, blk, 0, stream, p);
  else CMDA_LAUNCH((gemm_glds_kernel<TM, TN, AKS, BKS, ACONV, BCONV, NW, NSV, false>), grid, blk, 0, stream, p);
  CMDA_CHECK_LAUNCH();
}

template <int TM, int TN, int NW, int NSV>
int launch_glds_ns(const GemmParams& p, const dim3& grid, void* stream) {
  const bool aks = p.a_kstrided != 0, bks = p.b_kstrided != 0, ac = p.A.conv == 1, bc = p.B.conv == 1;  // (2 = patch view: plain fills)
  if (!aks && !bks && !ac && !bc) return launch_glds_mode<TM, TN, NW, NSV, false, false, false, false>(p, grid, stream);
  if (!aks && bks && !ac && !bc) return launch_glds_mode<TM, TN, NW, NSV, false, true, false, false>(p, grid, stream);
  if (aks && bks && !ac && !bc) return launch_glds_mode<TM, TN, NW, NSV, true, true, false, false>(p, grid, stream);
  if (aks && !bks && !ac && !bc) return launch_glds_mode<TM, TN, NW, NSV, true, false, false, false>(p, grid, stream);
  if (!aks && !bks && ac && !bc) return launch_glds_mode<TM, TN, NW, NSV, false, false, true, false>(p, grid, stream);
  if (aks && bks && !ac && bc) return launch_glds_mode<TM, TN, NW, NSV, true, true, false, true>(p, grid, stream);
  return CMDA_ERR_UNSUPPORTED;
}

template <int TM, int TN, int NW = 4>
int launch_glds(const GemmParams& p, void* stream) {
  constexpr int BM = 16 * TM * (NW / 2), BN = 32 * TN;
  const long tiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const long zz = (long)p.batch * p.batch2 * p.splits;
  if (tiles > 0x7fffffffL || zz > 65535) return CMDA_ERR_SHAPE;
  dim3 grid((unsigned)tiles, 1, (unsigned)zz);
  if constexpr (NW == 4) {
    // latency configuration: the whole grid is resident at once even at 4 stages (64x64: 64 KiB -> 2 blocks per CU; the wider
    // tiles: 96 / 128 KiB -> 1 block per CU) and every block runs >= 12 k-tiles (K >= 768)
    constexpr long kStage = (long)(BM + BN) * 128;
    const long resident = 256L * (4 * kStage <= 65536 ? 2 : 1);
    const long nkt = ((long)p.K + 63) / 64 / (p.splits > 0 ? p.splits : 1);
    const int force = p.tile_hint > 0 ? ((p.tile_hint >> 4) & 15) : 0;   // tuning sweeps (tools/gemm_sweep.py): stages in bits 4.. of the hint
    if (force) return force == 4 ? launch_glds_ns<TM, TN, NW, 4>(p, grid, stream) : launch_glds_ns<TM, TN, NW, 0>(p, grid, stream);
    // (deeper than 4 -- 8 x 16 KiB on the 64x64 tile, 6 x 24 KiB on 128x64, one block per CU -- measured slower on every shape
    // of the step: profiles/r02_gemm_sweep.txt)
    const int min_nkt = (p.tile_hint > 0 && (p.tile_hint >> 12)) ? (p.tile_hint >> 12) : 12;   // (bits 12..: tuning sweeps; in the step 12 k-tiles measured
                                                                                                // 77.5-78.1 ms against 78.0-78.2 at 3 and 78.2-79.2 at 40)
    if (tiles * zz <= resident && nkt >= min_nkt) return launch_glds_ns<TM, TN, NW, 4>(p, grid, stream);
  }
  return launch_glds_ns<TM, TN, NW, 0>(p, grid, stream);
}

// grouped launch of the weight-gradient operand modes (A and B K-strided; B plain / patch view, or an im2col view)
// (NW = 8 on the same tiles measured SLOWER on the UDA step, 58.9 -> 59.2 ms, gpurun r04v: these grids are many waves of blocks deep,
// the four-wave blocks' co-residency hides the DMA issue rate that binds a lone block)
template <int TM, int TN, int NW = 4>
int launch_glds_grouped(const GemmParams* tab, const void* blk, int nblocks, int bconv, void* stream) {
  if (nblocks <= 0) return CMDA_OK;
  const dim3 grid((unsigned)nblocks), blkdim(64 * NW);
  const int* b = reinterpret_cast<const int*>(blk);
  // bconv = operand class of the bucket (gemm_grouped.hip): 0 every problem plain with K % 64 == 0, 1 plain / patch views, 2 im2col B
  if (bconv == 2) CMDA_LAUNCH((gemm_glds_grouped_kernel<TM, TN, true, true, false, true, NW, 0, false>), grid, blkdim, 0, stream, tab, b);
  else if (bconv == 1) CMDA_LAUNCH((gemm_glds_grouped_kernel<TM, TN, true, true, false, false, NW, 0, false>), grid, blkdim, 0, stream, tab, b);
  else CMDA_LAUNCH((gemm_glds_grouped_kernel<TM, TN, true, true, false, false, NW, 0, true>), grid, blkdim, 0, stream, tab, b);
  CMDA_CHECK_LAUNCH();
}

template <typename T, int TM, int TN>
int launch_tile(const GemmParams& p, void* stream) {
  constexpr int BM = 32 * TM, BN = 32 * TN;
  const long tiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  if (tiles > 0x7fffffffL || (long)p.batch * p.batch2 * p.splits > 65535) return CMDA_ERR_SHAPE;
  dim3 grid((unsigned)tiles, 1, (unsigned)(p.batch * p.batch2 * p.splits));
  const bool aks = p.a_kstrided != 0, bks = p.b_kstrided != 0;
  if (!aks && !bks) CMDA_LAUNCH((gemm_kernel<T, TM, TN, false, false>), grid, dim3(256), 0, stream, p);
  else if (!aks && bks) CMDA_LAUNCH((gemm_kernel<T, TM, TN, false, true>), grid, dim3(256), 0, stream, p);
  else if (aks && bks) CMDA_LAUNCH((gemm_kernel<T, TM, TN, true, true>), grid, dim3(256), 0, stream, p);
  else CMDA_LAUNCH((gemm_kernel<T, TM, TN, true, false>), grid, dim3(256), 0, stream, p);
  CMDA_CHECK_LAUNCH();
}

}  // namespace

// cross-unit entry points (C++ linkage, not part of the C ABI)
int cmda_gemm_glds_t0_(const cmda_gemm_params_t& p, void* stream);        // gemm_t0.hip: 128x128 tile
int cmda_gemm_glds_t1_(const cmda_gemm_params_t& p, void* stream);        // gemm_t1.hip: 128x64 tile
int cmda_gemm_glds_t2_(const cmda_gemm_params_t& p, void* stream);        // gemm_t2.hip: 64x64 tile
int cmda_gemm_glds_t3_(const cmda_gemm_params_t& p, void* stream);        // gemm_t3.hip: 256x256 tile, 8 waves
bool cmda_gemm_lean_ok_(const cmda_gemm_params_t& p, int tile);           // gemm_lean.hip: lean plain-operand instance (64x64 / 128x64)
int cmda_gemm_lean_(const cmda_gemm_params_t& p, int tile, int four_stage, void* stream);
int cmda_gemm_grouped_t2_(const cmda_gemm_params_t* tab, const void* blk, int nblocks, int bconv, void* stream);  // gemm_g2.hip: 64x64
int cmda_gemm_grouped_t0_(const cmda_gemm_params_t* tab, const void* blk, int nblocks, int bconv, void* stream);  // gemm_g0.hip: 128x128
int cmda_gemm_grouped_t1_(const cmda_gemm_params_t* tab, const void* blk, int nblocks, int bconv, void* stream);  // gemm_g1.hip: 128x64
int cmda_gemm_grouped_t3_(const cmda_gemm_params_t* tab, const void* blk, int nblocks, int bconv, void* stream);  // gemm_g3.hip: 64x128
int cmda_gemm_wg_grouped_(const cmda_gemm_params_t* tab, const void* blk, int nblocks, int bconv, void* stream);  // gemm_wg.hip, grouped
int cmda_gemm_wg_(const cmda_gemm_params_t& p, void* stream);             // gemm_wg.hip: 256x256 weight-gradient kernel (32x32x16 MFMA, atomics)
int cmda_gemm_pp_(const cmda_gemm_params_t& p, void* stream);             // gemm_pp.hip: 256x256 ping-pong kernel (32x32x16 MFMA)
int cmda_gemm_reg_(const cmda_gemm_params_t& p, int tile, void* stream);  // gemm_reg.hip: register-staged kernels, dispatch
int cmda_gemm_x3_(const cmda_gemm_params_t& p, int tile, void* stream);
bool cmda_gemm_x3_lean_ok_(const cmda_gemm_params_t& p);                   // gemm_x3_lean.hip: LDS-DMA split-bf16 kernel, plain operands
int cmda_gemm_x3_lean_(const cmda_gemm_params_t& p, void* stream);
int cmda_gemm_x3_lean_grouped_(const cmda_gemm_params_t* tab, const void* blk, int nblocks, void* stream);   // grouped weight-gradient form   // gemm_x3.hip: fp32 storage, split-bf16 (bf16 x 3) MFMA
// gemm_reg_{f32,bf16}_t{0,1,2}.hip: one (dtype, tile) each -- these are the slow units to compile (~50 s apiece)
int cmda_gemm_reg_f32_t0_(const cmda_gemm_params_t& p, void* stream);
int cmda_gemm_reg_f32_t1_(const cmda_gemm_params_t& p, void* stream);
int cmda_gemm_reg_f32_t2_(const cmda_gemm_params_t& p, void* stream);
int cmda_gemm_reg_bf16_t0_(const cmda_gemm_params_t& p, void* stream);
int cmda_gemm_reg_bf16_t1_(const cmda_gemm_params_t& p, void* stream);
int cmda_gemm_reg_bf16_t2_(const cmda_gemm_params_t& p, void* stream);
