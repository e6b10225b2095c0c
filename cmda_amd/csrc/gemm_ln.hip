// gemm_ln.hip -- LayerNorm in the PROLOGUE of the Linear behind it: norm1 -> q and attn.norm -> kv of a MiT block
// (mix_transformer.py:86-92,123-139: x + attn(norm1(x)); kv = self.kv(self.norm(self.sr(x)))).  At 2 + 2 samples per GPU each of
// the two LayerNorms is a 3 - 4 us kernel plus a dependent-launch boundary in front of a 6 us GEMM; here the GEMM's workgroup
// normalises its own 64-row panel of the fp32 input on the way into LDS (the whole K = C <= 384 extent of the panel stays resident:
// 64 x 320 x 2 B = 40 KiB) and runs the lean kernel's k-loop with only the weight tiles coming through the LDS-DMA ring.
// Every n-tile's workgroup repeats the normalisation of its rows (N / 64 = 5 ... 10 times per row, L2 hits); the workgroups of
// n-tile 0 also store the normalised rows and the statistics (the spatial-reduction convolution / the weight gradient / the
// LayerNorm backward read them).
// Arithmetic = ln_fwd_kernel's (layernorm.hip): two-pass mean / variance in fp32, the same bf16 rounding; the sums associate differently
// (eight lanes per row), so rows and statistics agree with the separate launch to fp32 round-off, not bit for bit.
#include "gemm_kernels.h"

namespace {

struct LnGemmParams {
  const void* x;          // [M, K] input rows (fp32 or bf16), contiguous
  const float* gamma;
  const float* beta;
  bf16_t* xn;             // [M, K] normalised rows (bf16) or null
  float* mean;            // [M] or null
  float* rstd;
  const bf16_t* B;        // [N, K] weights, K-contiguous
  void* C;
  const float* bias;
  long ldb, ldc;
  int M, N, K, tiles_n;
  int ntile, lpr, act, flags;   // flags: 1 out_f32, 4 c_vec_ok
  float alpha, eps;
};

// 64 x 64 tile on eight waves (the lean kernel's 2-stage configuration); NVL = K / 32 16-byte vectors per lane (compile-time: every
// load of the row is issued unconditionally, back to back -- a run-time bound put each one behind its own branch and wait)
template <typename TX, int NVL>
__global__ __launch_bounds__(512, 1) void gemm_ln_kernel(LnGemmParams q) {
  typedef bf16_t T;
  constexpr int NW = 8, NT = 64 * NW, BM = 64, BN = 64, BK = 64, TN = 2;
  constexpr int SZ_A = BM * BK, SZ_B = BN * BK, PITCH_C = BN + 4;
  CMDA_DYN_SMEM(smem);    // [K / 64][64 rows][64] normalised panel, then two weight stages
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, g = lane >> 4, l15 = lane & 15;
  const int M = q.M, N = q.N, K = q.K, nkt = K >> 6, tiles_n = q.tiles_n, ntile = q.ntile;
  T* const sA = reinterpret_cast<T*>(smem);
  T* const sBbase = sA + nkt * SZ_A;
  int bt = blockIdx.x;
  {   // XCD-contiguous tile ranges (gemm_lean_kernel)
    const int qq = ntile >> 3, rr = ntile & 7, xcd = bt & 7, loc = bt >> 3;
    bt = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + loc;
  }
  const int mt = (int)((unsigned)bt / (unsigned)tiles_n), nt = bt - mt * tiles_n;
  const long m0 = (long)mt * BM, n0 = (long)nt * BN;

  // weight tiles: one DMA instruction per wave per k-tile (line = 8 * wid + lane / 8, the LDS image of gemm_glds_body)
  const char* curB;
  int stepB;
  {
    const int ln = wid * 8 + (lane >> 3), chunk = (lane & 7) ^ (ln & 7);
    const long r = n0 + ln;
    const bool ok = r < N;
    curB = ok ? reinterpret_cast<const char*>(q.B + r * q.ldb + chunk * 8) : reinterpret_cast<const char*>(g_zero16);
    stepB = ok ? BK * 2 : 0;
  }
  auto issue = [&](int stage) {
    glds16(curB, reinterpret_cast<char*>(sBbase + stage * SZ_B) + wid * 1024);
    curB += stepB;
  };
  issue(0);

  // ---- LayerNorm of the panel's rows: EIGHT lanes per row (thread t: row t / 8, 16-byte vectors t % 8, t % 8 + 8, ...), so the 512
  // threads cover the 64 rows at once, every load is in flight before the first reduction and a reduction is three xor-shuffles.
  // (First version: ln_fwd_kernel's grouping, a wave per row for C = 320, eight rows per wave one after the other -- bit-identical to
  // the separate launch but 4 us SLOWER than it: every n-tile repeats the rows' shuffle chains.  The sums here associate differently:
  // the normalised rows agree with ln_fwd_kernel's to fp32 round-off before the bf16 rounding.)
  {
    constexpr int MAXV = NVL;
    constexpr int nvl = NVL;
    const int l8 = tid & 7, prow = tid >> 3;
    const long row = m0 + prow;
    const bool live = row < M;
    const TX* xr = reinterpret_cast<const TX*>(q.x) + (live ? row : (long)M - 1) * K;   // (rows past M: read the last row, store nothing)
    const bool store = nt == 0 && live;
    float v[MAXV][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) ld4(xr + (i * 8 + l8) * 4, v[i]);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    const float mean = s / (float)K;
    float qs = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      if (i < nvl) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = v[i][j] - mean;
          qs += d * d;
        }
      }
    }
    qs += __shfl_xor(qs, 1, 64);
    qs += __shfl_xor(qs, 2, 64);
    qs += __shfl_xor(qs, 4, 64);
    const float rstd = rsqrtf(qs / (float)K + q.eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      if (i < nvl) {
        const int vi = i * 8 + l8;
        float gm[4], bt4[4], o[4];
        ld4(q.gamma + vi * 4, gm);
        ld4(q.beta + vi * 4, bt4);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = live ? (v[i][j] - mean) * rstd * gm[j] + bt4[j] : 0.f;
        // column 4 vi of the row: k-tile vi / 16, 16-byte chunk (vi % 16) / 2, half vi % 2 (swizzled like the DMA image)
        T* dst = sA + (vi >> 4) * SZ_A + prow * BK + (((((vi & 15) >> 1)) ^ (prow & 7)) << 3) + ((vi & 1) << 2);
        st4(dst, o);
        if (store && q.xn) st4(q.xn + row * K + vi * 4, o);
      }
    }
    if (store && l8 == 0 && q.mean) {
      q.mean[row] = mean;
      q.rstd[row] = rstd;
    }
  }

  // epilogue operands requested now (gemm_lean_kernel)
  constexpr int QPR = BN / 4, RSTEP = NT / QPR, NIT = BM / RSTEP;
  const int q4 = (tid % QPR) * 4, er0 = tid / QPR;
  const long en = n0 + q4;
  const bool ecol = en < N;
  const bool full = (q.flags & 4) != 0 && en + 4 <= N;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (q.bias && ecol) {
    if (full) ld4(q.bias + en, bv);
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (en + e < N) bv[e] = q.bias[en + e];
    }
  }

  f32x4 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int st = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    pipe_barrier<0>();     // weight tile kt has landed (and, at kt = 0, every wave's part of the panel is written); stage st ^ 1 is free
    if (kt + 1 < nkt) issue(st ^ 1);
    const T* sAk = sA + kt * SZ_A;
    const T* sB = sBbase + st * SZ_B;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int row = wm * 16 + l15;
      const u16x8 fa = *reinterpret_cast<const u16x8*>(&sAk[row * BK + (((kk * 4 + g) ^ (row & 7)) << 3)]);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int rb = wn * 16 * TN + j * 16 + l15;
        const u16x8 fb = *reinterpret_cast<const u16x8*>(&sB[rb * BK + (((kk * 4 + g) ^ (rb & 7)) << 3)]);
        acc[j] = mfma_bf16_16x16x32(fa, fb, acc[j]);
      }
    }
    st ^= 1;
  }
  __syncthreads();
  float* sC = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) sC[(wm * 16 + 4 * g + r) * PITCH_C + wn * 16 * TN + j * 16 + l15] = acc[j][r];
  __syncthreads();
  if (!ecol) return;
  const float alpha = q.alpha;
  const bool f32o = (q.flags & 1) != 0;
  const int act = q.act;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int row = er0 + it * RSTEP;
    const long m = m0 + row;
    if (m >= M) break;
    const float4 t = *reinterpret_cast<const float4*>(&sC[row * PITCH_C + q4]);
    float v[4] = {t.x, t.y, t.z, t.w};
    const long ci = m * q.ldc + en;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = alpha * v[e] + bv[e];
      v[e] = act == 0 ? x : act == 1 ? epi_act<1>(x) : act == 2 ? epi_act<2>(x) : epi_act<3>(x);
    }
    if (full) {
      if (f32o) st4(reinterpret_cast<float*>(q.C) + ci, v);
      else st4(reinterpret_cast<T*>(q.C) + ci, v);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (en + e >= N) continue;
        if (f32o) reinterpret_cast<float*>(q.C)[ci + e] = v[e];
        else stf(reinterpret_cast<T*>(q.C) + ci + e, v[e]);
      }
    }
  }
}

}  // namespace

// y = act(LayerNorm(x) W^T + b) with the normalised rows / statistics as side outputs.  `p`: the Linear as a cmda_gemm problem whose
// A view is the normalised-row buffer xn (bf16 [M, K], contiguous; may not be null: the fallback needs it).  Problems the fused kernel
// does not take (other dtypes, K > 384, epilogues with a residual, ...) run as cmda_layernorm_fwd2 + cmda_gemm: same results.
extern "C" int cmda_ln_gemm(const cmda_gemm_params_t* pp, const void* x, int x_dtype, const float* gamma, const float* beta, float eps,
                            float* mean, float* rstd, int store_xn, void* stream) {
  if (!pp || !x || !gamma || !beta) return CMDA_ERR_SHAPE;
  const cmda_gemm_params_t& p = *pp;
  if (p.M <= 0) return CMDA_OK;
  auto plain = [](const GemmView& v) { return v.conv == 0 && v.vec_ok && (v.ld % 8) == 0 && v.R < (1L << 31) && v.Cc < (1L << 31); };
  const int K = p.K;
  const bool fused = p.dtype == CMDA_BF16 && (x_dtype == CMDA_F32 || x_dtype == CMDA_BF16) && !p.a_kstrided && !p.b_kstrided && plain(p.A) &&
                     plain(p.B) && p.A.ld == K && (K % 64) == 0 && K >= 64 && K <= 384 && p.batch == 1 && p.batch2 <= 1 && p.splits <= 1 &&
                     !p.atomic && !p.colsum && p.c_patch_ow == 0 && p.c_perm_ci == 0 && !p.res && !p.rowscale && p.beta == 0.f &&
                     (mean == nullptr) == (rstd == nullptr) && !(p.tile_hint > 0 && (p.tile_hint & 8192)) && p.A.ptr != nullptr &&
                     (reinterpret_cast<uintptr_t>(x) % 16) == 0 && (reinterpret_cast<uintptr_t>(p.A.ptr) % 16) == 0 &&
                     // measured (tools/dbg/ln_gemm_bench.py, us per dependent launch group, fused against LayerNorm + Linear):
                     //   65536 x 64 x 64  9.9 / 16.6    16384 x 128 x 128  7.6 / 10.8    2048 x 320 x 320  7.8 / 8.7    1024 x 640 x 320  7.8 / 8.3
                     //   4096 x 320 x 320  11.8 / 10.4 -- a fused workgroup is one serial chain (row loads -> statistics -> LDS -> k-loop, 5.7 us
                     //   against the lean kernel's 2.3) and a grid of more than one workgroup per CU pays it twice; short rows stay ahead
                     (K <= 128 || (long)((p.M + 63) / 64) * ((p.N + 63) / 64) <= 256 || (p.tile_hint > 0 && (p.tile_hint & 32768)));
  if (!fused) {
    const int rc = cmda_layernorm_fwd2(x, x_dtype, gamma, beta, const_cast<void*>(p.A.ptr), p.dtype == CMDA_BF16 ? CMDA_BF16 : CMDA_F32, mean,
                                       rstd, p.M, K, eps, stream);
    if (rc != CMDA_OK) return rc;
    return cmda_gemm(pp, stream);
  }
  LnGemmParams q;
  q.x = x; q.gamma = gamma; q.beta = beta;
  q.xn = store_xn ? reinterpret_cast<bf16_t*>(const_cast<void*>(p.A.ptr)) : nullptr;
  q.mean = mean; q.rstd = rstd;
  q.B = reinterpret_cast<const bf16_t*>(p.B.ptr);
  q.C = p.C; q.bias = p.bias;
  q.ldb = p.B.ld; q.ldc = p.ldc;
  q.M = p.M; q.N = p.N; q.K = K;
  q.tiles_n = (p.N + 63) / 64;
  const long tiles = (long)((p.M + 63) / 64) * q.tiles_n;
  if (tiles > 0x7fffffffL) return CMDA_ERR_SHAPE;
  q.ntile = (int)tiles;
  q.lpr = 8;
  q.act = p.act;
  q.flags = (p.out_f32 ? 1 : 0) | (p.c_vec_ok ? 4 : 0);
  q.alpha = p.alpha; q.eps = eps;
  const size_t lds = (size_t)(K / 64) * 64 * 64 * 2 + 2 * 64 * 64 * 2;
  const dim3 grid((unsigned)tiles), blk(512);
#define CMDA_LN_GEMM_T(TX)                                                                          \
  switch (K >> 5) {                                                                                 \
    case 2: CMDA_LAUNCH((gemm_ln_kernel<TX, 2>), grid, blk, lds, stream, q); break;                  \
    case 4: CMDA_LAUNCH((gemm_ln_kernel<TX, 4>), grid, blk, lds, stream, q); break;                  \
    case 6: CMDA_LAUNCH((gemm_ln_kernel<TX, 6>), grid, blk, lds, stream, q); break;                  \
    case 8: CMDA_LAUNCH((gemm_ln_kernel<TX, 8>), grid, blk, lds, stream, q); break;                  \
    case 10: CMDA_LAUNCH((gemm_ln_kernel<TX, 10>), grid, blk, lds, stream, q); break;                \
    default: CMDA_LAUNCH((gemm_ln_kernel<TX, 12>), grid, blk, lds, stream, q); break;                \
  }
  if (x_dtype == CMDA_F32) { CMDA_LN_GEMM_T(float) } else { CMDA_LN_GEMM_T(bf16_t) }
#undef CMDA_LN_GEMM_T
  CMDA_CHECK_LAUNCH();
}
