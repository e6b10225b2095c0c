// attention.hip -- fused spatial-reduction attention core (bf16, head_dim 64, up to 256 keys): scores, softmax and the
// weighted sum of values in ONE kernel each way; the [N, Nk] score / probability matrices never reach HBM.
//
// Reference ops replaced: Attention.forward of mmseg/models/backbones/mix_transformer.py:86-103
//   attn = (q @ k.transpose(-2, -1)) * scale ; attn = attn.softmax(dim=-1) ; x = (attn @ v)      (attn_drop = 0)
// and its autograd backward.  q comes from the `q` Linear ([B*N, C], head h = columns 64h..64h+63), k / v from the `kv`
// Linear over the spatially reduced tokens ([B*Nk, 2C]: K at column 64h, V at column C + 64h).  With CMDA's 512x512
// crops every stage has Nk = 256 keys (sr_ratios 8/4/2/1), which is what makes the whole key set fit in LDS.
//
// Orientation: everything is computed TRANSPOSED, S^T = K Q^T, so that the MFMA result layout (row = 4*(lane>>4)+r,
// col = lane&15) puts the keys on (lane>>4, r) and the queries on lane&15.  Then
//   * the softmax over keys is a per-lane reduction plus two xor-shuffles (16, 32),
//   * P^T is already in the B-operand layout of the second GEMM O^T = V^T P^T (k = keys): no LDS round trip for P,
//   * V^T is read from the row-major V tile with ds_read_b64_tr_b16.
// The k index of an MFMA is a free labelling as long as A and B agree; here slot (g, j) of the u-th group of 32 keys is
// key 32u + 4g + j (j < 4) or 32u + 16 + 4g + (j - 4), i.e. exactly the accumulator registers of score tiles 2u and 2u+1.
//
// MFMA-bound per block but small: per 16 queries 32 + 32 v_mfma_f32_16x16x32_bf16.  HBM traffic per (batch, head):
// Q, O once, K, V once per 128-query block (L2 hits).  Algorithmic bytes per query: 2 * 64 * 2 (q, o) + the K/V share.
#include "common.h"

namespace {

constexpr int kHD = 64;     // head dim
constexpr int kMaxK = 256;  // keys kept in LDS (forward and backward)
constexpr int kMaxKFwd = 320;  // ... by the forward-only instance
constexpr int kNT = kMaxK / 16;

struct AttnParams {
  const bf16_t* q;   // [B*N, C]
  const bf16_t* kv;  // [B*Nk, 2C]
  bf16_t* o;         // [B*N, C]
  int B, N, Nk, heads, C;
  float scale;
  int q_per_block;
};

// XCD-aware block order: workgroups are dealt round-robin to the 8 XCDs (one L2 each).  All query blocks of one
// (batch, head) read the same K / V, so the 1-D grid is remapped to give each XCD a contiguous band of logical blocks
// (query block fastest): a (batch, head)'s K / V then comes through ONE L2 instead of eight (PMC: 67 MB fetched per
// launch against 17 MB written before this).
static __device__ __forceinline__ unsigned xcd_logical_block() {
  const unsigned nb = gridDim.x, lin = blockIdx.x;
  const unsigned q = nb / 8, r = nb % 8, xcd = lin % 8, loc = lin / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// K or V of one (batch, head) -> LDS [kMaxK][64] bf16, 128-byte lines, 16-byte chunk c of line r stored at slot c ^ (r & 7);
// rows >= Nk come from the zero block.  One DMA instruction moves 8 lines (1 KiB); the 4 waves take 8 instructions each.
extern __device__ __attribute__((aligned(16))) unsigned g_attn_zero16[4];
__device__ __attribute__((aligned(16))) unsigned g_attn_zero16[4] = {0u, 0u, 0u, 0u};

template <int MAXK = kMaxK>
static __device__ __forceinline__ void load_kv_tile(const bf16_t* __restrict__ src, int ld, int Nk, bf16_t* lds, int wid,
                                                    int lane, int nwaves) {
  for (int i = wid; i < MAXK / 8; i += nwaves) {
    const int row = 8 * i + (lane >> 3);
    const int chunk = (lane & 7) ^ (row & 7);
    const void* s = row < Nk ? static_cast<const void*>(src + (long)row * ld + chunk * 8)
                             : static_cast<const void*>(g_attn_zero16);
    glds16(s, reinterpret_cast<char*>(lds) + i * 1024);
  }
}

// A operand (rows = 16 consecutive tile rows, k = 32 consecutive d) of a row-major swizzled [rows][64] tile
static __device__ __forceinline__ u16x8 frag_rows(const bf16_t* tile, int row0, int kk, int g, int l15) {
  const int row = row0 + l15;
  return *reinterpret_cast<const u16x8*>(&tile[row * kHD + (((kk * 4 + g) ^ (row & 7)) << 3)]);
}

// A operand of the transposed tile: rows = d (16dt + l15), k slot (g, j) = tile row R0 + j (j < 4) / R1 + (j - 4)
static __device__ __forceinline__ u16x8 frag_cols(const bf16_t* tile, int R0, int R1, int dt, int l15) {
  const int q = l15 >> 2, pp = l15 & 3;
  const int r0 = R0 + q, r1 = R1 + q;
  const int cidx = 2 * dt + (pp >> 1), half = (pp & 1) << 2;
  const u16x4 lo = lds_read_tr16(&tile[r0 * kHD + ((cidx ^ (r0 & 7)) << 3) + half]);
  const u16x4 hi = lds_read_tr16(&tile[r1 * kHD + ((cidx ^ (r1 & 7)) << 3) + half]);
  return u16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

static __device__ __forceinline__ float col_max(float v) {  // over the 4 lane groups holding one query column
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
static __device__ __forceinline__ float col_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// scores^T of 16 queries against all key tiles, then the softmax over keys: p[t][r] = P^T[key 16t + 4g + r][query l15]
template <int NT = kNT>
static __device__ __forceinline__ void scores_softmax(const bf16_t* sK, const u16x8 (&qf)[2], int nt, int Nk, float scale,
                                                      int g, int l15, f32x4 (&p)[NT], float* lse_out = nullptr) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    p[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t < nt) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) p[t] = mfma_bf16_16x16x32(frag_rows(sK, 16 * t, kk, g, l15), qf[kk], p[t]);
    }
  }
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool live = 16 * t + 4 * g + r < Nk;
      p[t][r] = live ? p[t][r] * scale : -INFINITY;
      m = fmaxf(m, p[t][r]);
    }
  m = col_max(m);
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[t][r] = __expf(p[t][r] - m);  // exp(-inf) = 0 for the masked keys
      l += p[t][r];
    }
  l = col_sum(l);
  if (lse_out) *lse_out = m + __logf(l);
  const float inv = 1.f / l;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) p[t][r] *= inv;
}

static __device__ __forceinline__ u16x8 pack_pair(const f32x4& a, const f32x4& b) {
  return u16x8{f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(b[0]), f2bf(b[1]), f2bf(b[2]), f2bf(b[3])};
}

// B operand b[k = d][col = query]: 16 bytes of row (row0 + l15) of a [rows, ld] global matrix; rows past `nrows` repeat the last
static __device__ __forceinline__ void load_qfrag(const bf16_t* __restrict__ base, long row0, long nrows, int ld, int g,
                                                  int l15, u16x8 (&f)[2]) {
  long row = row0 + l15;
  if (row >= nrows) row = nrows - 1;
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) f[kk] = *reinterpret_cast<const u16x8*>(base + row * ld + 32 * kk + 8 * g);
}

// queries per block of the forward-shaped kernels: 64 (one pass of 4 waves x 16) or 128 (two passes, K/V loaded once for
// both) -- the launcher takes 64 while that still leaves the grid under ~4 blocks per CU (stage 3: 27.4 -> 22.9 us)
static inline int fwd_queries_per_block(int B, int N, int heads) {
  return (long)((N + 127) / 128) * heads * B < 1024 ? 64 : 128;
}

// MAXK: keys held in LDS -- 256 (every stage of a 512 x 512 crop; two blocks per CU) or 320 (inference on 440 x 640 frames,
// encoder_decoder.py:897-936: 260 / 280 keys after the spatial reduction; forward only, one block per CU)
template <int MAXK>
__global__ __launch_bounds__(256, MAXK <= 256 ? 2 : 1) void attn_fwd_kernel(AttnParams p) {
  constexpr int NT = MAXK / 16;
  __shared__ __attribute__((aligned(1024))) bf16_t sK[MAXK * kHD];
  __shared__ __attribute__((aligned(1024))) bf16_t sV[MAXK * kHD];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const unsigned nqb = (unsigned)((p.N + p.q_per_block - 1) / p.q_per_block);
  const unsigned lb = xcd_logical_block(), bh = lb / nqb;
  const long qblk = lb - bh * nqb;
  const int h = (int)(bh % (unsigned)p.heads), b = (int)(bh / (unsigned)p.heads);
  const bf16_t* kbase = p.kv + (long)b * p.Nk * 2 * p.C + h * kHD;
  load_kv_tile<MAXK>(kbase, 2 * p.C, p.Nk, sK, wid, lane, 4);
  load_kv_tile<MAXK>(kbase + p.C, 2 * p.C, p.Nk, sV, wid, lane, 4);
  __syncthreads();
  const int nt = (p.Nk + 15) >> 4;
  const bf16_t* qb = p.q + (long)b * p.N * p.C + h * kHD;
  bf16_t* ob = p.o + (long)b * p.N * p.C + h * kHD;
  for (int pass = 0; pass < p.q_per_block / 64; ++pass) {
    const long q0 = qblk * p.q_per_block + pass * 64 + wid * 16;
    if (q0 >= p.N) break;  // wave-uniform
    u16x8 qf[2];
    load_qfrag(qb, q0, p.N, p.C, g, l15, qf);
    f32x4 pr[NT];
    scores_softmax<NT>(sK, qf, nt, p.Nk, p.scale, g, l15, pr);
    f32x4 oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NT / 2; ++u) {
      if (2 * u < nt) {
        const u16x8 pb = pack_pair(pr[2 * u], pr[2 * u + 1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          oacc[dt] = mfma_bf16_16x16x32(frag_cols(sV, 32 * u + 4 * g, 32 * u + 16 + 4 * g, dt, l15), pb, oacc[dt]);
      }
    }
    if (q0 + l15 < p.N) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const float v[4] = {oacc[dt][0], oacc[dt][1], oacc[dt][2], oacc[dt][3]};
        st4(ob + (q0 + l15) * p.C + 16 * dt + 4 * g, v);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward = two kernels, both recomputing the probabilities from q / k (nothing but q, kv, o-gradient is read):
//   attn_bwd_dq_kernel   (forward-shaped: 16 queries per wave, all keys in LDS)
//       P^T = softmax(K Q^T), dP^T = V dO^T, D = sum_k P dP, dS^T = scale * P (dP - D), dQ^T = K^T dS^T -> dq
//       and per query the softmax log-sum-exp and D -> stats[b, h, q, 2] for the second kernel
//   attn_bwd_dkv_kernel  (block = one 64-key slice of one (batch, head) x a span of queries; wave = 32 queries at a time)
//       S = Q K^T for the slice, P = exp(scale*S - lse), dP = dO V^T, dS = scale * P (dP - D)       [query][key] tiles
//       dV^T += dO^T P,  dK^T += Q^T dS      (contraction over the 32 queries; P / dS are B operands straight from the
//       accumulator registers, dO^T / Q^T are transposed LDS reads of the wave's staged 32 x 64 tiles)
//       waves fold their partial dK^T / dV^T through LDS, then one fp32 atomic per element per block.
// (The first version did all of it in one kernel with P / dS staged in LDS: 144 KiB of LDS and 512 registers meant one
// wave per SIMD and ~21 us per 64-query chunk -- latency-bound.  Split this way the kernels run 8 / 12 waves per CU.)
struct AttnBwdParams {
  const bf16_t* q;
  const bf16_t* kv;
  const bf16_t* d_o;
  bf16_t* dq;
  float* dkv32;
  bf16_t* dkv16;  // direct mode (one query span per key slice): dK | dV stored as bf16, no atomics, no workspace
  float* stats;  // [B, heads, N, 2] = (lse, D)
  int B, N, Nk, heads, C, q_per_block;
  float scale;
  int fwd_q_per_block;  // of the dQ kernel
  int spans;            // query spans of the dK/dV kernel
};

__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnBwdParams p) {
  __shared__ __attribute__((aligned(1024))) bf16_t sK[kMaxK * kHD];
  __shared__ __attribute__((aligned(1024))) bf16_t sV[kMaxK * kHD];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const unsigned nqb = (unsigned)((p.N + p.fwd_q_per_block - 1) / p.fwd_q_per_block);
  const unsigned lb = xcd_logical_block(), bh = lb / nqb;
  const long qblk = lb - bh * nqb;
  const int h = (int)(bh % (unsigned)p.heads), b = (int)(bh / (unsigned)p.heads);
  const bf16_t* kbase = p.kv + (long)b * p.Nk * 2 * p.C + h * kHD;
  load_kv_tile(kbase, 2 * p.C, p.Nk, sK, wid, lane, 4);
  load_kv_tile(kbase + p.C, 2 * p.C, p.Nk, sV, wid, lane, 4);
  __syncthreads();
  const int nt = (p.Nk + 15) >> 4;
  const long rowb = (long)b * p.N;
  const bf16_t* qb = p.q + rowb * p.C + h * kHD;
  const bf16_t* dob = p.d_o + rowb * p.C + h * kHD;
  bf16_t* dqb = p.dq + rowb * p.C + h * kHD;
  float* st = p.stats + ((long)b * p.heads + h) * p.N * 2;
  for (int pass = 0; pass < p.fwd_q_per_block / 64; ++pass) {
    const long q0 = qblk * p.fwd_q_per_block + pass * 64 + wid * 16;
    if (q0 >= p.N) break;  // wave-uniform
    u16x8 qf[2], dof[2];
    load_qfrag(qb, q0, p.N, p.C, g, l15, qf);
    load_qfrag(dob, q0, p.N, p.C, g, l15, dof);
    f32x4 pr[kNT], dp[kNT];
    float lse;
    scores_softmax(sK, qf, nt, p.Nk, p.scale, g, l15, pr, &lse);
    float dsum = 0.f;
#pragma unroll
    for (int t = 0; t < kNT; ++t) {
      dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (t < nt) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) dp[t] = mfma_bf16_16x16x32(frag_rows(sV, 16 * t, kk, g, l15), dof[kk], dp[t]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) dsum += pr[t][r] * dp[t][r];
    }
    dsum = col_sum(dsum);
#pragma unroll
    for (int t = 0; t < kNT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) dp[t][r] = p.scale * pr[t][r] * (dp[t][r] - dsum);  // dS^T
    f32x4 dqacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dqacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < kNT / 2; ++u) {
      if (2 * u < nt) {
        const u16x8 db = pack_pair(dp[2 * u], dp[2 * u + 1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          dqacc[dt] = mfma_bf16_16x16x32(frag_cols(sK, 32 * u + 4 * g, 32 * u + 16 + 4 * g, dt, l15), db, dqacc[dt]);
      }
    }
    if (q0 + l15 < p.N) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const float v[4] = {dqacc[dt][0], dqacc[dt][1], dqacc[dt][2], dqacc[dt][3]};
        st4(dqb + (q0 + l15) * p.C + 16 * dt + 4 * g, v);
      }
      if (g == 0) {
        st[(q0 + l15) * 2 + 0] = lse;
        st[(q0 + l15) * 2 + 1] = dsum;
      }
    }
  }
}

constexpr int kKS = 64;        // keys per dK/dV block
constexpr int kRedPitch = 68;  // floats per key row of the cross-wave reduction buffer

// NW waves per block: 4 (two blocks per CU when the queries are split into spans) or 8 (direct mode: one block per key slice walks
// every query, 80 ... 160 blocks per launch -- the walk is a serial chain of DMA -> MFMA steps per wave, so twice the waves halve it)
template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void attn_bwd_dkv_kernel(AttnBwdParams p) {
  __shared__ __attribute__((aligned(1024))) bf16_t sK[kKS * kHD];
  __shared__ __attribute__((aligned(1024))) bf16_t sV[kKS * kHD];
  constexpr int kStageBytes = NW * 2 * 32 * kHD * 2;          // per wave: Q and dO tiles of 32 queries
  constexpr int kRedBytes = 2 * kKS * kRedPitch * 4;          // dK | dV as [key][d] fp32
  __shared__ __attribute__((aligned(1024))) char sbuf[kStageBytes > kRedBytes ? kStageBytes : kRedBytes];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, l15 = lane & 15;
  // logical order: key slice fastest (the four slices of one (batch, head, span) stream the same Q / dO rows), then span
  const unsigned lb = xcd_logical_block();
  const int ks = (int)(lb & 3);
  const unsigned span = (lb >> 2) % (unsigned)p.spans, bh = (lb >> 2) / (unsigned)p.spans;
  const int h = (int)(bh % (unsigned)p.heads), b = (int)(bh / (unsigned)p.heads);
  const int key0 = ks * kKS;
  if (key0 >= p.Nk) return;  // block-uniform: this slice holds no key
  const int nkeys = min(kKS, p.Nk - key0);
  const bf16_t* kbase = p.kv + ((long)b * p.Nk + key0) * 2 * p.C + h * kHD;
  for (int i = wid; i < kKS / 8; i += NW) {
    const int row = 8 * i + (lane >> 3);
    const int chunk = (lane & 7) ^ (row & 7);
    const void* zero = static_cast<const void*>(g_attn_zero16);
    const bf16_t* src = kbase + (long)row * 2 * p.C + chunk * 8;
    glds16(row < nkeys ? static_cast<const void*>(src) : zero, reinterpret_cast<char*>(sK) + i * 1024);
    glds16(row < nkeys ? static_cast<const void*>(src + p.C) : zero, reinterpret_cast<char*>(sV) + i * 1024);
  }
  __syncthreads();
  const long rowb = (long)b * p.N;
  const bf16_t* qb = p.q + rowb * p.C + h * kHD;
  const bf16_t* dob = p.d_o + rowb * p.C + h * kHD;
  const float* st = p.stats + ((long)b * p.heads + h) * p.N * 2;
  bf16_t* sQw = reinterpret_cast<bf16_t*>(sbuf) + wid * 2 * 32 * kHD;
  bf16_t* sDOw = sQw + 32 * kHD;

  f32x4 dvacc[4][4], dkacc[4][4];  // [d tile][key tile]: dV^T, dK^T
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) dvacc[dt][kt] = dkacc[dt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const long qbeg = (long)span * p.q_per_block;
  const long qend = min((long)p.N, qbeg + p.q_per_block);
  for (long q32 = qbeg + 32 * wid; q32 < qend; q32 += 32 * NW) {
    // stage this wave's 32 rows of Q and dO (rows past N read as zero)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 8 * i + (lane >> 3);
      const int chunk = (lane & 7) ^ (row & 7);
      const bool ok = q32 + row < p.N;
      const void* zero = static_cast<const void*>(g_attn_zero16);
      glds16(ok ? static_cast<const void*>(qb + (q32 + row) * p.C + chunk * 8) : zero, reinterpret_cast<char*>(sQw) + i * 1024);
      glds16(ok ? static_cast<const void*>(dob + (q32 + row) * p.C + chunk * 8) : zero, reinterpret_cast<char*>(sDOw) + i * 1024);
    }
    float lse[2][4], dd[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long qi = q32 + 16 * qt + 4 * g + r;
        const bool ok = qi < p.N;
        lse[qt][r] = ok ? st[qi * 2] : INFINITY;  // exp(s - inf) = 0: a missing query contributes nothing
        dd[qt][r] = ok ? st[qi * 2 + 1] : 0.f;
      }
    dma_wait<0>();
    f32x4 pr[2][4], ds[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      u16x8 fq[2], fdo[2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        fq[kk] = frag_rows(sQw, 16 * qt, kk, g, l15);
        fdo[kk] = frag_rows(sDOw, 16 * qt, kk, g, l15);
      }
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          s = mfma_bf16_16x16x32(fq[kk], frag_rows(sK, 16 * kt, kk, g, l15), s);
          d = mfma_bf16_16x16x32(fdo[kk], frag_rows(sV, 16 * kt, kk, g, l15), d);
        }
        const bool live = 16 * kt + l15 < nkeys;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = live ? __expf(s[r] * p.scale - lse[qt][r]) : 0.f;
          pr[qt][kt][r] = pv;
          ds[qt][kt][r] = p.scale * pv * (d[r] - dd[qt][r]);
        }
      }
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const u16x8 fdoT = frag_cols(sDOw, 4 * g, 16 + 4 * g, dt, l15);
      const u16x8 fqT = frag_cols(sQw, 4 * g, 16 + 4 * g, dt, l15);
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        dvacc[dt][kt] = mfma_bf16_16x16x32(fdoT, pack_pair(pr[0][kt], pr[1][kt]), dvacc[dt][kt]);
        dkacc[dt][kt] = mfma_bf16_16x16x32(fqT, pack_pair(ds[0][kt], ds[1][kt]), dkacc[dt][kt]);
      }
    }
  }
  // ---- fold the four waves' partials through LDS ([key][d], pitch 68), then one atomic per element
  __syncthreads();  // every wave is done with its staging tiles (the buffer is reused)
  float* red = reinterpret_cast<float*>(sbuf);
  for (int w = 0; w < NW; ++w) {
    if (wid == w) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          float* rk = red + (16 * kt + l15) * kRedPitch + 16 * dt + 4 * g;
          float* rv = rk + kKS * kRedPitch;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            rk[r] = (w ? rk[r] : 0.f) + dkacc[dt][kt][r];
            rv[r] = (w ? rv[r] : 0.f) + dvacc[dt][kt][r];
          }
        }
    }
    __syncthreads();
  }
  if (p.dkv16) {  // this block saw every query of its (batch, head, key slice): the sums are final
    bf16_t* ob = p.dkv16 + ((long)b * p.Nk + key0) * 2 * p.C + h * kHD;
    for (int e = tid; e < kKS * kHD; e += 64 * NW) {
      const int kl = e >> 6, d = e & 63;
      if (kl < nkeys) {
        stf(ob + (long)kl * 2 * p.C + d, red[kl * kRedPitch + d]);
        stf(ob + (long)kl * 2 * p.C + p.C + d, red[(kKS + kl) * kRedPitch + d]);
      }
    }
    return;
  }
  float* ob = p.dkv32 + ((long)b * p.Nk + key0) * 2 * p.C + h * kHD;
  for (int e = tid; e < kKS * kHD; e += 64 * NW) {
    const int kl = e >> 6, d = e & 63;
    if (kl < nkeys) {
      atomicAdd(ob + (long)kl * 2 * p.C + d, red[kl * kRedPitch + d]);
      atomicAdd(ob + (long)kl * 2 * p.C + p.C + d, red[(kKS + kl) * kRedPitch + d]);
    }
  }
}


// ===============================================================================================================================
// SPLIT-bf16 instances (cmda dtype CMDA_F32X3: fp32 storage, every product as three bf16 MFMAs a_hi b_hi + a_hi b_lo + a_lo b_hi with
// fp32 accumulation, ~16 mantissa bits per product -- the arithmetic of gemm_x3_lean.hip): the tolerance-meeting mode's attention ran
// as batched Q K^T / softmax / P V GEMMs + four backward products + two softmax launches with the [N, Nk] probabilities in HBM
// (mix_transformer.py:97-101; ~20 ms of that mode's kernel time per UDA step).  Same orientation, same three kernels as above; what
// differs: q / kv / d_o arrive as fp32 and are split ONCE -- K and V into hi / lo bf16 images in LDS (4 x 32 KB: one workgroup per CU),
// the query-side fragments in registers --, the probabilities and dS are split in registers right where the bf16 kernels pack them, and
// the outputs are stored as fp32.
struct AttnX3Params {
  const float* q;
  const bf16_t* kvh;   // kv split ONCE in HBM by the caller (cmda_split_bf16): every query block of a (batch, head) reads the same K / V,
  const bf16_t* kvl;   // and converting the 128 KB per block in registers cost as much as the block's MFMAs (first version: 116.4 against 116.8 ms)
  float* o;
  int B, N, Nk, heads, C;
  float scale;
  int q_per_block;
};

static __device__ __forceinline__ void split4(const float (&x)[4], u16x4& hi, u16x4& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bf16_t h = f2bf(x[e]);
    hi[e] = h;
    lo[e] = f2bf(x[e] - bf2f(h));
  }
}

// this wave's LDS stores are visible to its own later LDS reads (cross-lane hand-off inside ONE wave: DS operations of a wave execute in
// order; the statement also keeps the compiler from moving the reads up)
#ifndef CMDA_EMU
static __device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#else
static inline void wave_lds_sync() { emu::wave_barrier(); }
#endif

// rows [0, rows) of an fp32 [*, ld] matrix (64 columns at src) -> hi / lo images [rows][64] bf16 in the swizzled layout of load_kv_tile
// (16-byte chunk c of line r at slot c ^ (r & 7)); rows >= nvalid are zero.  Thread t of nthr converts one float4 per step.
static __device__ __forceinline__ void load_split_tile(const float* __restrict__ src, long ld, int nvalid, int rows, bf16_t* hi, bf16_t* lo,
                                                       int t, int nthr) {
  for (int i = t; i < rows * 16; i += nthr) {
    const int row = i >> 4, c4 = i & 15;
    float x[4] = {0.f, 0.f, 0.f, 0.f};
    if (row < nvalid) ld4(src + (long)row * ld + 4 * c4, x);
    u16x4 h, l;
    split4(x, h, l);
    const int off = row * kHD + ((((c4 >> 1) ^ (row & 7))) << 3) + ((c4 & 1) << 2);
    *reinterpret_cast<u16x4*>(hi + off) = h;
    *reinterpret_cast<u16x4*>(lo + off) = l;
  }
}

// B operand b[k = d][col = query] of 16 consecutive rows of an fp32 [rows, ld] matrix, split: lane (l15, g) reads d = 32 kk + 8 g .. + 7
static __device__ __forceinline__ void load_qfrag_x3(const float* __restrict__ base, long row0, long nrows, int ld, int g, int l15,
                                                     u16x8 (&fh)[2], u16x8 (&fl)[2]) {
  long row = row0 + l15;
  if (row >= nrows) row = nrows - 1;
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    float a[4], b[4];
    ld4(base + row * ld + 32 * kk + 8 * g, a);
    ld4(base + row * ld + 32 * kk + 8 * g + 4, b);
    u16x4 ah, al, bh, bl;
    split4(a, ah, al);
    split4(b, bh, bl);
    fh[kk] = u16x8{ah[0], ah[1], ah[2], ah[3], bh[0], bh[1], bh[2], bh[3]};
    fl[kk] = u16x8{al[0], al[1], al[2], al[3], bl[0], bl[1], bl[2], bl[3]};
  }
}

// c += a b with a = (ah, al), b = (bh, bl): the three significant partial products
static __device__ __forceinline__ f32x4 mfma_x3(u16x8 ah, u16x8 al, u16x8 bh, u16x8 bl, f32x4 c) {
  c = mfma_bf16_16x16x32(al, bh, c);
  c = mfma_bf16_16x16x32(ah, bl, c);
  return mfma_bf16_16x16x32(ah, bh, c);
}

static __device__ __forceinline__ void pack_pair_x3(const f32x4& a, const f32x4& b, u16x8& hi, u16x8& lo) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const bf16_t ha = f2bf(a[r]), hb = f2bf(b[r]);
    hi[r] = ha; hi[4 + r] = hb;
    lo[r] = f2bf(a[r] - bf2f(ha)); lo[4 + r] = f2bf(b[r] - bf2f(hb));
  }
}

// scores^T of 16 queries against all key tiles (three products per tile and k half), then the softmax over keys -- scores_softmax
static __device__ __forceinline__ void scores_softmax_x3(const bf16_t* sKh, const bf16_t* sKl, const u16x8 (&qh)[2], const u16x8 (&ql)[2],
                                                         int nt, int Nk, float scale, int g, int l15, f32x4 (&p)[kNT], float* lse_out = nullptr) {
#pragma unroll
  for (int t = 0; t < kNT; ++t) {
    p[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t < nt) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        p[t] = mfma_x3(frag_rows(sKh, 16 * t, kk, g, l15), frag_rows(sKl, 16 * t, kk, g, l15), qh[kk], ql[kk], p[t]);
    }
  }
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < kNT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool live = 16 * t + 4 * g + r < Nk;
      p[t][r] = live ? p[t][r] * scale : -INFINITY;
      m = fmaxf(m, p[t][r]);
    }
  m = col_max(m);
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < kNT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[t][r] = __expf(p[t][r] - m);
      l += p[t][r];
    }
  l = col_sum(l);
  if (lse_out) *lse_out = m + __logf(l);
  const float inv = 1.f / l;
#pragma unroll
  for (int t = 0; t < kNT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) p[t][r] *= inv;
}

__global__ __launch_bounds__(256, 1) void attn_fwd_x3_kernel(AttnX3Params p) {
  __shared__ __attribute__((aligned(1024))) bf16_t sKh[kMaxK * kHD];
  __shared__ __attribute__((aligned(1024))) bf16_t sKl[kMaxK * kHD];
  __shared__ __attribute__((aligned(1024))) bf16_t sVh[kMaxK * kHD];
  __shared__ __attribute__((aligned(1024))) bf16_t sVl[kMaxK * kHD];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const unsigned nqb = (unsigned)((p.N + p.q_per_block - 1) / p.q_per_block);
  const unsigned lb = xcd_logical_block(), bh = lb / nqb;
  const long qblk = lb - bh * nqb;
  const int h = (int)(bh % (unsigned)p.heads), b = (int)(bh / (unsigned)p.heads);
  const long koff = (long)b * p.Nk * 2 * p.C + h * kHD;
  const int nt = (p.Nk + 15) >> 4;
  load_kv_tile(p.kvh + koff, 2 * p.C, p.Nk, sKh, wid, lane, 4);
  load_kv_tile(p.kvl + koff, 2 * p.C, p.Nk, sKl, wid, lane, 4);
  load_kv_tile(p.kvh + koff + p.C, 2 * p.C, p.Nk, sVh, wid, lane, 4);
  load_kv_tile(p.kvl + koff + p.C, 2 * p.C, p.Nk, sVl, wid, lane, 4);
  __syncthreads();
  const float* qb = p.q + (long)b * p.N * p.C + h * kHD;
  float* ob = p.o + (long)b * p.N * p.C + h * kHD;
  for (int pass = 0; pass < p.q_per_block / 64; ++pass) {
    const long q0 = qblk * p.q_per_block + pass * 64 + wid * 16;
    if (q0 >= p.N) break;  // wave-uniform
    u16x8 qh[2], ql[2];
    load_qfrag_x3(qb, q0, p.N, p.C, g, l15, qh, ql);
    f32x4 pr[kNT];
    scores_softmax_x3(sKh, sKl, qh, ql, nt, p.Nk, p.scale, g, l15, pr);
    f32x4 oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < kNT / 2; ++u) {
      if (2 * u < nt) {
        u16x8 ph, pl;
        pack_pair_x3(pr[2 * u], pr[2 * u + 1], ph, pl);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          oacc[dt] = mfma_x3(frag_cols(sVh, 32 * u + 4 * g, 32 * u + 16 + 4 * g, dt, l15), frag_cols(sVl, 32 * u + 4 * g, 32 * u + 16 + 4 * g, dt, l15),
                             ph, pl, oacc[dt]);
      }
    }
    if (q0 + l15 < p.N) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const float v[4] = {oacc[dt][0], oacc[dt][1], oacc[dt][2], oacc[dt][3]};
        st4(ob + (q0 + l15) * p.C + 16 * dt + 4 * g, v);
      }
    }
  }
}

struct AttnBwdX3Params {
  const float* q;
  const bf16_t* kvh;
  const bf16_t* kvl;
  const float* d_o;
  float* dq;
  float* dkv32;    // accumulate mode: fp32 atomics (zero on entry)
  float* dkv_out;  // direct mode: dK | dV stored
  float* stats;
  int B, N, Nk, heads, C, q_per_block;
  float scale;
  int fwd_q_per_block;
  int spans;
};

__global__ __launch_bounds__(256, 1) void attn_bwd_dq_x3_kernel(AttnBwdX3Params p) {
  __shared__ __attribute__((aligned(1024))) bf16_t sKh[kMaxK * kHD];
  __shared__ __attribute__((aligned(1024))) bf16_t sKl[kMaxK * kHD];
  __shared__ __attribute__((aligned(1024))) bf16_t sVh[kMaxK * kHD];
  __shared__ __attribute__((aligned(1024))) bf16_t sVl[kMaxK * kHD];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const unsigned nqb = (unsigned)((p.N + p.fwd_q_per_block - 1) / p.fwd_q_per_block);
  const unsigned lb = xcd_logical_block(), bh = lb / nqb;
  const long qblk = lb - bh * nqb;
  const int h = (int)(bh % (unsigned)p.heads), b = (int)(bh / (unsigned)p.heads);
  const long koff = (long)b * p.Nk * 2 * p.C + h * kHD;
  const int nt = (p.Nk + 15) >> 4;
  load_kv_tile(p.kvh + koff, 2 * p.C, p.Nk, sKh, wid, lane, 4);
  load_kv_tile(p.kvl + koff, 2 * p.C, p.Nk, sKl, wid, lane, 4);
  load_kv_tile(p.kvh + koff + p.C, 2 * p.C, p.Nk, sVh, wid, lane, 4);
  load_kv_tile(p.kvl + koff + p.C, 2 * p.C, p.Nk, sVl, wid, lane, 4);
  __syncthreads();
  const long rowb = (long)b * p.N;
  const float* qb = p.q + rowb * p.C + h * kHD;
  const float* dob = p.d_o + rowb * p.C + h * kHD;
  float* dqb = p.dq + rowb * p.C + h * kHD;
  float* st = p.stats + ((long)b * p.heads + h) * p.N * 2;
  for (int pass = 0; pass < p.fwd_q_per_block / 64; ++pass) {
    const long q0 = qblk * p.fwd_q_per_block + pass * 64 + wid * 16;
    if (q0 >= p.N) break;  // wave-uniform
    u16x8 qh[2], ql[2], doh[2], dol[2];
    load_qfrag_x3(qb, q0, p.N, p.C, g, l15, qh, ql);
    load_qfrag_x3(dob, q0, p.N, p.C, g, l15, doh, dol);
    f32x4 pr[kNT], dp[kNT];
    float lse;
    scores_softmax_x3(sKh, sKl, qh, ql, nt, p.Nk, p.scale, g, l15, pr, &lse);
    float dsum = 0.f;
#pragma unroll
    for (int t = 0; t < kNT; ++t) {
      dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (t < nt) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          dp[t] = mfma_x3(frag_rows(sVh, 16 * t, kk, g, l15), frag_rows(sVl, 16 * t, kk, g, l15), doh[kk], dol[kk], dp[t]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) dsum += pr[t][r] * dp[t][r];
    }
    dsum = col_sum(dsum);
#pragma unroll
    for (int t = 0; t < kNT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) dp[t][r] = p.scale * pr[t][r] * (dp[t][r] - dsum);  // dS^T
    f32x4 dqacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dqacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < kNT / 2; ++u) {
      if (2 * u < nt) {
        u16x8 dh, dl;
        pack_pair_x3(dp[2 * u], dp[2 * u + 1], dh, dl);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          dqacc[dt] = mfma_x3(frag_cols(sKh, 32 * u + 4 * g, 32 * u + 16 + 4 * g, dt, l15), frag_cols(sKl, 32 * u + 4 * g, 32 * u + 16 + 4 * g, dt, l15),
                              dh, dl, dqacc[dt]);
      }
    }
    if (q0 + l15 < p.N) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const float v[4] = {dqacc[dt][0], dqacc[dt][1], dqacc[dt][2], dqacc[dt][3]};
        st4(dqb + (q0 + l15) * p.C + 16 * dt + 4 * g, v);
      }
      if (g == 0) {
        st[(q0 + l15) * 2 + 0] = lse;
        st[(q0 + l15) * 2 + 1] = dsum;
      }
    }
  }
}

// dK | dV, split-bf16: four waves per 64-key slice; per wave the 32 x 64 tiles of Q and dO as hi / lo images (4 x 4 KB)
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_x3_kernel(AttnBwdX3Params p) {
  constexpr int NW = 4;
  __shared__ __attribute__((aligned(1024))) bf16_t sKh[kKS * kHD];
  __shared__ __attribute__((aligned(1024))) bf16_t sKl[kKS * kHD];
  __shared__ __attribute__((aligned(1024))) bf16_t sVh[kKS * kHD];
  __shared__ __attribute__((aligned(1024))) bf16_t sVl[kKS * kHD];
  constexpr int kStageBytes = NW * 4 * 32 * kHD * 2;          // per wave: Q hi / lo and dO hi / lo tiles of 32 queries
  constexpr int kRedBytes = 2 * kKS * kRedPitch * 4;          // dK | dV as [key][d] fp32
  __shared__ __attribute__((aligned(1024))) char sbuf[kStageBytes > kRedBytes ? kStageBytes : kRedBytes];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const unsigned lb = xcd_logical_block();
  const int ks = (int)(lb & 3);
  const unsigned span = (lb >> 2) % (unsigned)p.spans, bh = (lb >> 2) / (unsigned)p.spans;
  const int h = (int)(bh % (unsigned)p.heads), b = (int)(bh / (unsigned)p.heads);
  const int key0 = ks * kKS;
  if (key0 >= p.Nk) return;  // block-uniform: this slice holds no key
  const int nkeys = min(kKS, p.Nk - key0);
  const long koff = ((long)b * p.Nk + key0) * 2 * p.C + h * kHD;
  for (int i = wid; i < kKS / 8; i += NW) {
    const int row = 8 * i + (lane >> 3);
    const int chunk = (lane & 7) ^ (row & 7);
    const void* zero = static_cast<const void*>(g_attn_zero16);
    const long so = koff + (long)row * 2 * p.C + chunk * 8;
    const bool ok = row < nkeys;
    glds16(ok ? static_cast<const void*>(p.kvh + so) : zero, reinterpret_cast<char*>(sKh) + i * 1024);
    glds16(ok ? static_cast<const void*>(p.kvl + so) : zero, reinterpret_cast<char*>(sKl) + i * 1024);
    glds16(ok ? static_cast<const void*>(p.kvh + so + p.C) : zero, reinterpret_cast<char*>(sVh) + i * 1024);
    glds16(ok ? static_cast<const void*>(p.kvl + so + p.C) : zero, reinterpret_cast<char*>(sVl) + i * 1024);
  }
  __syncthreads();
  const long rowb = (long)b * p.N;
  const float* qb = p.q + rowb * p.C + h * kHD;
  const float* dob = p.d_o + rowb * p.C + h * kHD;
  const float* st = p.stats + ((long)b * p.heads + h) * p.N * 2;
  bf16_t* sQh = reinterpret_cast<bf16_t*>(sbuf) + wid * 4 * 32 * kHD;
  bf16_t* sQl = sQh + 32 * kHD;
  bf16_t* sDh = sQl + 32 * kHD;
  bf16_t* sDl = sDh + 32 * kHD;

  f32x4 dvacc[4][4], dkacc[4][4];  // [d tile][key tile]: dV^T, dK^T
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) dvacc[dt][kt] = dkacc[dt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const long qbeg = (long)span * p.q_per_block;
  const long qend = min((long)p.N, qbeg + p.q_per_block);
  for (long q32 = qbeg + 32 * wid; q32 < qend; q32 += 32 * NW) {
    wave_lds_sync();   // the previous trip's fragment reads are done before the tiles are overwritten
    const int nq = (int)min(32L, p.N - q32);
    load_split_tile(qb + q32 * p.C, p.C, nq, 32, sQh, sQl, lane, 64);
    load_split_tile(dob + q32 * p.C, p.C, nq, 32, sDh, sDl, lane, 64);
    float lse[2][4], dd[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long qi = q32 + 16 * qt + 4 * g + r;
        const bool ok = qi < p.N;
        lse[qt][r] = ok ? st[qi * 2] : INFINITY;  // exp(s - inf) = 0: a missing query contributes nothing
        dd[qt][r] = ok ? st[qi * 2 + 1] : 0.f;
      }
    wave_lds_sync();
    f32x4 pr[2][4], ds[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      u16x8 fqh[2], fql[2], fdh[2], fdl[2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        fqh[kk] = frag_rows(sQh, 16 * qt, kk, g, l15);
        fql[kk] = frag_rows(sQl, 16 * qt, kk, g, l15);
        fdh[kk] = frag_rows(sDh, 16 * qt, kk, g, l15);
        fdl[kk] = frag_rows(sDl, 16 * qt, kk, g, l15);
      }
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          s = mfma_x3(fqh[kk], fql[kk], frag_rows(sKh, 16 * kt, kk, g, l15), frag_rows(sKl, 16 * kt, kk, g, l15), s);
          d = mfma_x3(fdh[kk], fdl[kk], frag_rows(sVh, 16 * kt, kk, g, l15), frag_rows(sVl, 16 * kt, kk, g, l15), d);
        }
        const bool live = 16 * kt + l15 < nkeys;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = live ? __expf(s[r] * p.scale - lse[qt][r]) : 0.f;
          pr[qt][kt][r] = pv;
          ds[qt][kt][r] = p.scale * pv * (d[r] - dd[qt][r]);
        }
      }
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const u16x8 fdoTh = frag_cols(sDh, 4 * g, 16 + 4 * g, dt, l15), fdoTl = frag_cols(sDl, 4 * g, 16 + 4 * g, dt, l15);
      const u16x8 fqTh = frag_cols(sQh, 4 * g, 16 + 4 * g, dt, l15), fqTl = frag_cols(sQl, 4 * g, 16 + 4 * g, dt, l15);
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        u16x8 ph, pl, dh, dl;
        pack_pair_x3(pr[0][kt], pr[1][kt], ph, pl);
        pack_pair_x3(ds[0][kt], ds[1][kt], dh, dl);
        dvacc[dt][kt] = mfma_x3(fdoTh, fdoTl, ph, pl, dvacc[dt][kt]);
        dkacc[dt][kt] = mfma_x3(fqTh, fqTl, dh, dl, dkacc[dt][kt]);
      }
    }
  }
  // ---- fold the four waves' partials through LDS ([key][d], pitch 68), then store (direct mode) or one atomic per element
  __syncthreads();  // every wave is done with its staging tiles (the buffer is reused)
  float* red = reinterpret_cast<float*>(sbuf);
  for (int w = 0; w < NW; ++w) {
    if (wid == w) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          float* rk = red + (16 * kt + l15) * kRedPitch + 16 * dt + 4 * g;
          float* rv = rk + kKS * kRedPitch;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            rk[r] = (w ? rk[r] : 0.f) + dkacc[dt][kt][r];
            rv[r] = (w ? rv[r] : 0.f) + dvacc[dt][kt][r];
          }
        }
    }
    __syncthreads();
  }
  if (p.dkv_out) {
    float* ob = p.dkv_out + ((long)b * p.Nk + key0) * 2 * p.C + h * kHD;
    for (int e = tid; e < kKS * kHD; e += 64 * NW) {
      const int kl = e >> 6, d = e & 63;
      if (kl < nkeys) {
        ob[(long)kl * 2 * p.C + d] = red[kl * kRedPitch + d];
        ob[(long)kl * 2 * p.C + p.C + d] = red[(kKS + kl) * kRedPitch + d];
      }
    }
    return;
  }
  float* ob = p.dkv32 + ((long)b * p.Nk + key0) * 2 * p.C + h * kHD;
  for (int e = tid; e < kKS * kHD; e += 64 * NW) {
    const int kl = e >> 6, d = e & 63;
    if (kl < nkeys) {
      atomicAdd(ob + (long)kl * 2 * p.C + d, red[kl * kRedPitch + d]);
      atomicAdd(ob + (long)kl * 2 * p.C + p.C + d, red[(kKS + kl) * kRedPitch + d]);
    }
  }
}

}  // namespace

// q [B*N, C] bf16, kv [B*Nk, 2C] bf16 -> o [B*N, C] bf16; head_dim = C / heads must be 64, Nk <= 320 (the backward: 256), C % 8 == 0.
extern "C" int cmda_attention_fwd(const void* q, const void* kv, void* o, int B, int N, int Nk, int heads, int C,
                                  float scale, int dtype, void* stream) {
  if (B <= 0 || N <= 0) return CMDA_OK;
  if (dtype != CMDA_BF16) return CMDA_ERR_DTYPE;
  if (heads <= 0 || C != heads * kHD || Nk <= 0 || Nk > kMaxKFwd) return CMDA_ERR_UNSUPPORTED;
  if (heads > 65535 || B > 65535) return CMDA_ERR_SHAPE;
  const int qpb = fwd_queries_per_block(B, N, heads);
  AttnParams p{(const bf16_t*)q, (const bf16_t*)kv, (bf16_t*)o, B, N, Nk, heads, C, scale, qpb};
  const long nblk = (long)((N + qpb - 1) / qpb) * heads * B;
  if (nblk > 0x7fffffffL) return CMDA_ERR_SHAPE;
  dim3 grid((unsigned)nblk);
  if (Nk <= kMaxK) CMDA_LAUNCH(attn_fwd_kernel<kMaxK>, grid, dim3(256), 0, stream, p);
  else CMDA_LAUNCH(attn_fwd_kernel<kMaxKFwd>, grid, dim3(256), 0, stream, p);
  CMDA_CHECK_LAUNCH();
}

// d_o [B*N, C] bf16 -> dq [B*N, C] bf16 (written), dkv32 [B*Nk, 2C] fp32 (accumulated: dK | dV).
// stats: scratch of cmda_attention_bwd_ws_floats(B, N, heads) floats.
extern "C" int64_t cmda_attention_bwd_ws_floats(int B, int N, int heads) { return (int64_t)B * N * heads * 2; }

// direct mode: with at most 1024 queries per (batch, head) ONE block per 64-key slice walks all of them, so dK | dV need no
// cross-block accumulation: they are stored as bf16 into dkv16 (no fp32 workspace, no atomics -- ~12 of the kernel's ~24 us at the
// stage-3 shape -- and no cast pass afterwards).  1 = cmda_attention_bwd will take dkv16 for these sizes.
extern "C" int cmda_attention_bwd_direct(int B, int N, int Nk, int heads) {
  (void)B; (void)heads;
  return N <= 1024 && Nk <= kMaxK ? 1 : 0;
}

extern "C" int cmda_attention_bwd(const void* q, const void* kv, const void* d_o, void* dq, float* dkv32, void* dkv16,
                                  float* stats, int B, int N, int Nk, int heads, int C, float scale, int dtype, void* stream) {
  if (B <= 0 || N <= 0) return CMDA_OK;
  if (dtype != CMDA_BF16) return CMDA_ERR_DTYPE;
  if (heads <= 0 || C != heads * kHD || Nk <= 0 || Nk > kMaxK) return CMDA_ERR_UNSUPPORTED;
  if (heads * 4 > 65535 || B > 65535) return CMDA_ERR_SHAPE;
  const bool direct = cmda_attention_bwd_direct(B, N, Nk, heads) != 0 && dkv16 != nullptr;
  if (!direct && dkv32 == nullptr) return CMDA_ERR_SHAPE;
  const int fqpb = fwd_queries_per_block(B, N, heads);
  AttnBwdParams p{(const bf16_t*)q, (const bf16_t*)kv, (const bf16_t*)d_o, (bf16_t*)dq, dkv32, direct ? (bf16_t*)dkv16 : nullptr, stats,
                  B, N, Nk, heads, C, 0, scale, fqpb};
  const long nblk1 = (long)((N + fqpb - 1) / fqpb) * heads * B;
  if (nblk1 > 0x7fffffffL) return CMDA_ERR_SHAPE;
  dim3 g1((unsigned)nblk1);
  CMDA_LAUNCH(attn_bwd_dq_kernel, g1, dim3(256), 0, stream, p);
  // dK/dV: (batch, head, key slice) x query spans, spans a multiple of 128 queries.  Every span costs one fp32 atomic per
  // dK/dV element (~1.3 TB/s chip-wide: 1280 blocks of the stage-3 shape spent 31 of their 61 us there), so only as many
  // spans as it takes to reach ~2 blocks per CU -- and a single span (direct mode) when the queries are few.
  const long slices = (long)B * heads * ((Nk + kKS - 1) / kKS);
  long spans = direct ? 1 : std::max<long>(1, 512 / slices);
  long qpb = ((N + spans - 1) / spans + 127) / 128 * 128;
  spans = (N + qpb - 1) / qpb;
  p.q_per_block = (int)qpb;
  p.spans = (int)spans;
  dim3 g2((unsigned)(spans * heads * 4 * B));
  if (direct && N > 512) CMDA_LAUNCH(attn_bwd_dkv_kernel<8>, g2, dim3(512), 0, stream, p);   // (N = 256: 21.4 us against 20.4 with 4 waves)
  else CMDA_LAUNCH(attn_bwd_dkv_kernel<4>, g2, dim3(256), 0, stream, p);
  CMDA_CHECK_LAUNCH();
}

// ---- split-bf16 instances (fp32 storage): q / o / d_o / dq / dK | dV fp32; kv_hi / kv_lo = cmda_split_bf16 of the fp32 [B*Nk, 2C] kv
extern "C" int cmda_attention_fwd_x3(const float* q, const void* kv_hi, const void* kv_lo, float* o, int B, int N, int Nk, int heads, int C,
                                     float scale, void* stream) {
  if (B <= 0 || N <= 0) return CMDA_OK;
  if (heads <= 0 || C != heads * kHD || Nk <= 0 || Nk > kMaxK || (C & 3)) return CMDA_ERR_UNSUPPORTED;
  if (heads > 65535 || B > 65535) return CMDA_ERR_SHAPE;
  const int qpb = (long)((N + 127) / 128) * heads * B < 256 ? 64 : 128;   // one workgroup per CU (128 KB of K / V images): 64 queries while the grid is under two rounds
  AttnX3Params p{q, (const bf16_t*)kv_hi, (const bf16_t*)kv_lo, o, B, N, Nk, heads, C, scale, qpb};
  const long nblk = (long)((N + qpb - 1) / qpb) * heads * B;
  if (nblk > 0x7fffffffL) return CMDA_ERR_SHAPE;
  CMDA_LAUNCH(attn_fwd_x3_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, p);
  CMDA_CHECK_LAUNCH();
}

extern "C" int cmda_attention_bwd_x3(const float* q, const void* kv_hi, const void* kv_lo, const float* d_o, float* dq, float* dkv32,
                                     float* dkv_direct, float* stats, int B, int N, int Nk, int heads, int C, float scale, void* stream) {
  if (B <= 0 || N <= 0) return CMDA_OK;
  if (heads <= 0 || C != heads * kHD || Nk <= 0 || Nk > kMaxK || (C & 3)) return CMDA_ERR_UNSUPPORTED;
  if (heads * 4 > 65535 || B > 65535) return CMDA_ERR_SHAPE;
  const bool direct = cmda_attention_bwd_direct(B, N, Nk, heads) != 0 && dkv_direct != nullptr;
  if (!direct && dkv32 == nullptr) return CMDA_ERR_SHAPE;
  const int fq = (long)((N + 127) / 128) * heads * B < 256 ? 64 : 128;
  AttnBwdX3Params p{q, (const bf16_t*)kv_hi, (const bf16_t*)kv_lo, d_o, dq, dkv32, direct ? dkv_direct : nullptr, stats,
                    B, N, Nk, heads, C, 0, scale, fq, 1};
  const long nb1 = (long)((N + fq - 1) / fq) * heads * B;
  if (nb1 > 0x7fffffffL) return CMDA_ERR_SHAPE;
  CMDA_LAUNCH(attn_bwd_dq_x3_kernel, dim3((unsigned)nb1), dim3(256), 0, stream, p);
  const long slices = (long)B * heads * ((Nk + kKS - 1) / kKS);
  long spans = direct ? 1 : std::max<long>(1, 256 / slices);
  long qpb = ((N + spans - 1) / spans + 127) / 128 * 128;
  spans = (N + qpb - 1) / qpb;
  p.q_per_block = (int)qpb;
  p.spans = (int)spans;
  CMDA_LAUNCH(attn_bwd_dkv_x3_kernel, dim3((unsigned)(spans * heads * 4 * B)), dim3(256), 0, stream, p);
  CMDA_CHECK_LAUNCH();
}

