// gemm_lean2.hip -- second lean instance of the LDS-DMA GEMM for the encoders' Linear layers and their data gradients
// (q / kv / proj / fc1 / fc2 of mix_transformer.py:31-44,62-66,80-102 and the dgrads torch autograd would run for them), built on
// what the phase stamps of gemm_lean.hip showed (profiles/r05_lean_phase.txt):
//   * SWAPPED-operand MFMA: the W fragment goes in as the matrix instruction's A operand and the activation fragment as its B operand, so
//     a lane's four accumulator registers are four CONSECUTIVE output columns of ONE output row -- the epilogue adds bias / residual and
//     stores 8 (bf16) or 16 (fp32) bytes per lane straight from registers: no accumulator round trip through LDS, no epilogue barrier;
//   * tiles that fit the GRID: BM = 16 TM WM, BN = 16 TN WN for any TN (64 x 80, 128 x 80, 128 x 160, 256 x 160);
//   * optional PRODUCER waves (NWP > 0): waves NWC.. only issue the LDS-DMA pieces, the NWC consumer waves only read fragments and issue MFMAs;
//   * NS-stage ring with counted vmcnt waits (pieces per wave may differ by one: the wait is chosen per wave).
// MEASURED SLOWER than gemm_lean.hip on every encoder shape (profiles/r06_lean2_sweep.txt) and removed from the tree in the next commit;
// kept in the history as the experiment that was measured.  Dispatch hook (gemm.hip, inside `if (nt_plain && !no_glds)`):
//   if (p.tile_hint > 0 && (p.tile_hint & 32768) && cmda_gemm_lean2_ok_(p)) return cmda_gemm_lean2_(p, (p.tile_hint >> 16) & 63, stream);
#include "gemm_kernels.h"

namespace {

struct Lean2Params {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const float* bias;
  const void* res;
  const float* rowscale;
  long lda, ldb, ldc, ldres;
  int M, N, nkt, tiles_n;
  int ntile, rows_per_scale, act, flags;   // flags: 1 out_f32, 2 res_f32
  float alpha, beta;
  long p_img;
  int p_ohw, p_ow, p_row, p_col, p_seg, p_jump;   // patch view of A (kernel == stride convolution): see gemm_lean.hip
};

template <int TILES, int PPW>
static __device__ __forceinline__ void wait_tiles(int n) {
  if constexpr (TILES == 0) dma_wait<0>();
  else {
    if (n == PPW) dma_wait<TILES * PPW>();
    else dma_wait<TILES * (PPW > 1 ? PPW - 1 : 0)>();
  }
}

template <int TM, int TN, int WM, int WN, int NWP, int NS, bool PATCH>
__global__ __launch_bounds__(64 * (WM * WN + NWP), (WM * WN + NWP + 3) / 4) void gemm_lean2_kernel(Lean2Params q) {
  typedef bf16_t T;
  constexpr int NWC = WM * WN, NI = NWP ? NWP : NWC;
  constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN, BK = 64;
  constexpr int SZ_A = BM * BK, SZ_B = BN * BK;
  constexpr int PA = BM / 8, PB = BN / 8, PT = PA + PB;  // 1-KiB DMA pieces per k-tile (8 rows x 128 bytes each)
  constexpr int PPW = (PT + NI - 1) / NI;
  static_assert(PT >= NI, "fewer DMA pieces than issuing waves");
  __shared__ __attribute__((aligned(1024))) char smem[(size_t)NS * (SZ_A + SZ_B) * sizeof(T)];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const bool issuer = NWP ? wid >= NWC : true;
  const bool consumer = wid < NWC;
  const int M = q.M, N = q.N, nkt = q.nkt, tiles_n = q.tiles_n, ntile = q.ntile;
  int bt = blockIdx.x;
  {
    const int qq = ntile >> 3, rr = ntile & 7, xcd = bt & 7, loc = bt >> 3;
    bt = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + loc;
  }
  const int mt = (int)((unsigned)bt / (unsigned)tiles_n), nt = bt - mt * tiles_n;
  const long m0 = (long)mt * BM, n0 = (long)nt * BN;
  const char* cur[PPW];
  int step[PPW], jump[PPW];
  int ldsb[PPW], ldss[PPW];
  int my_n = 0;
  if (issuer) {
    const int ii = NWP ? wid - NWC : wid;
    const char* zero = reinterpret_cast<const char*>(g_zero16);
    const long lda = q.lda, ldb = q.ldb;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int p = ii + j * NI;
      cur[j] = zero; step[j] = 0; jump[j] = 0; ldsb[j] = 0; ldss[j] = 0;
      if (p < PT) {
        ++my_n;
        if (p < PA) {
          const int ln = p * 8 + (lane >> 3), chunk = (lane & 7) ^ (ln & 7);
          ldsb[j] = p * 1024; ldss[j] = SZ_A * 2;
          if constexpr (!PATCH) {
            const long r = m0 + ln;
            if (r < M) { cur[j] = reinterpret_cast<const char*>(q.A + r * lda + chunk * 8); step[j] = BK * 2; }
          } else {
            const unsigned r = (unsigned)min(m0 + ln, (long)M - 1);
            const unsigned b = r / (unsigned)q.p_ohw, rem = r - b * (unsigned)q.p_ohw, oh = rem / (unsigned)q.p_ow, ow = rem - oh * (unsigned)q.p_ow;
            cur[j] = reinterpret_cast<const char*>(q.A + (long)b * q.p_img + (long)oh * q.p_row + (long)ow * q.p_col + chunk * 8);
            step[j] = BK * 2; jump[j] = q.p_jump;
          }
        } else {
          const int pb = p - PA;
          const int ln = pb * 8 + (lane >> 3), chunk = (lane & 7) ^ (ln & 7);
          ldsb[j] = NS * SZ_A * 2 + pb * 1024; ldss[j] = SZ_B * 2;
          const long r = n0 + ln;
          if (r < N) { cur[j] = reinterpret_cast<const char*>(q.B + r * ldb + chunk * 8); step[j] = BK * 2; }
        }
      }
    }
  }
  int seg_left = PATCH ? q.p_seg : 0;
  auto issue = [&](int stage) {
#pragma unroll
    for (int j = 0; j < PPW; ++j)
      if (j < my_n) {
        glds16(cur[j], smem + ldsb[j] + stage * ldss[j]);
        cur[j] += step[j];
      }
    if constexpr (PATCH) {
      if (--seg_left == 0) {
        seg_left = q.p_seg;
#pragma unroll
        for (int j = 0; j < PPW; ++j) cur[j] += jump[j];
      }
    }
  };
  if (issuer) {
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
      if (s < nkt) issue(s);
  }
  const int wm = wid / WN, wn = wid - wm * WN;
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const T* const sAbase = reinterpret_cast<const T*>(smem);
  const T* const sBbase = sAbase + NS * SZ_A;
  int st = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    if (issuer) {
      if (kt + NS - 2 < nkt) wait_tiles<NS - 2, PPW>(my_n);
      else dma_wait<0>();
    }
    lds_barrier();
    if (issuer && kt + NS - 1 < nkt) {
      int sn = st + NS - 1;
      if (sn >= NS) sn -= NS;
      issue(sn);
    }
    if (consumer) {
      const T* sA = sAbase + st * SZ_A;
      const T* sB = sBbase + st * SZ_B;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        u16x8 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = (wm * TM + i) * 16 + l15;
          fa[i] = *reinterpret_cast<const u16x8*>(&sA[row * BK + (((kk * 4 + g) ^ (row & 7)) << 3)]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = (wn * TN + j) * 16 + l15;
          fb[j] = *reinterpret_cast<const u16x8*>(&sB[row * BK + (((kk * 4 + g) ^ (row & 7)) << 3)]);
        }
        // swapped operands: D[n-index 4g + r][m-index l15] -- the lane holds C[m0 + .. + l15][n0 + .. + 4g + r], r = 0..3
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = mfma_bf16_16x16x32(fb[j], fa[i], acc[i][j]);
      }
    }
    if (++st == NS) st = 0;
  }
  if (!consumer) return;
  const bool f32o = (q.flags & 1) != 0, res32 = (q.flags & 2) != 0;
  const bool has_res = q.res != nullptr, has_rs = q.rowscale != nullptr;
  const float alpha = q.alpha, beta = q.beta;
  const bool has_beta = beta != 0.f;
  const int act = q.act;
  const long nb = n0 + (long)wn * TN * 16 + 4 * g;
  float bv[TN][4];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    bv[j][0] = bv[j][1] = bv[j][2] = bv[j][3] = 0.f;
    if (q.bias && nb + j * 16 < N) ld4(q.bias + nb + j * 16, bv[j]);
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long m = m0 + (long)(wm * TM + i) * 16 + l15;
    if (m >= M) continue;
    float rs = 1.f;
    if (has_rs) rs = q.rowscale[(unsigned)m / (unsigned)q.rows_per_scale];
    float rv[TN][4], ov[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const long n = nb + j * 16;
      rv[j][0] = rv[j][1] = rv[j][2] = rv[j][3] = 0.f;
      ov[j][0] = ov[j][1] = ov[j][2] = ov[j][3] = 0.f;
      if (n >= N) continue;
      if (has_res) {
        if (res32) ld4(reinterpret_cast<const float*>(q.res) + m * q.ldres + n, rv[j]);
        else ld4(reinterpret_cast<const T*>(q.res) + m * q.ldres + n, rv[j]);
      }
      if (has_beta) {
        if (f32o) ld4(reinterpret_cast<const float*>(q.C) + m * q.ldc + n, ov[j]);
        else ld4(reinterpret_cast<const T*>(q.C) + m * q.ldc + n, ov[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const long n = nb + j * 16;
      if (n >= N) continue;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = alpha * acc[i][j][e] + bv[j][e];
        const float a = act == 0 ? x : act == 1 ? epi_act<1>(x) : act == 2 ? epi_act<2>(x) : epi_act<3>(x);
        v[e] = a * rs + rv[j][e] + beta * ov[j][e];
      }
      if (f32o) st4(reinterpret_cast<float*>(q.C) + m * q.ldc + n, v);
      else st4(reinterpret_cast<T*>(q.C) + m * q.ldc + n, v);
    }
  }
}

template <int TM, int TN, int WM, int WN, int NWP, int NS>
int launch_lean2(const GemmParams& p, void* stream) {
  constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN;
  Lean2Params q;
  q.A = reinterpret_cast<const bf16_t*>(p.A.ptr);
  q.B = reinterpret_cast<const bf16_t*>(p.B.ptr);
  q.C = p.C;
  q.bias = p.bias;
  q.res = p.res;
  q.rowscale = p.rowscale;
  q.lda = p.A.ld; q.ldb = p.B.ld; q.ldc = p.ldc; q.ldres = p.ldres;
  q.M = p.M; q.N = p.N; q.nkt = p.K / 64;
  q.tiles_n = (p.N + BN - 1) / BN;
  const long tiles = (long)((p.M + BM - 1) / BM) * q.tiles_n;
  if (tiles > 0x7fffffffL) return CMDA_ERR_SHAPE;
  q.ntile = (int)tiles;
  q.rows_per_scale = p.rows_per_scale > 0 ? p.rows_per_scale : 1;
  q.act = p.act;
  q.flags = (p.out_f32 ? 1 : 0) | (p.res_f32 ? 2 : 0);
  q.alpha = p.alpha; q.beta = p.beta;
  q.p_img = 0; q.p_ohw = q.p_ow = 1; q.p_row = q.p_col = q.p_seg = q.p_jump = 0;
  const dim3 grid((unsigned)tiles), blk(64 * (WM * WN + NWP));
  if (p.A.conv == 2) {
    const GemmView& v = p.A;
    q.p_img = (long)v.H * v.W * v.C;
    q.p_ohw = v.OH * v.OW; q.p_ow = v.OW;
    q.p_row = v.stride * v.W * v.C; q.p_col = v.stride * v.C;
    q.p_seg = v.KW * v.C / 64;
    q.p_jump = (v.W - v.KW) * v.C * 2;
    CMDA_LAUNCH((gemm_lean2_kernel<TM, TN, WM, WN, NWP, NS, true>), grid, blk, 0, stream, q);
    CMDA_CHECK_LAUNCH();
  }
  CMDA_LAUNCH((gemm_lean2_kernel<TM, TN, WM, WN, NWP, NS, false>), grid, blk, 0, stream, q);
  CMDA_CHECK_LAUNCH();
}

}  // namespace

bool cmda_gemm_lean2_ok_(const cmda_gemm_params_t& p) {
  auto plain = [](const GemmView& v) { return v.conv == 0 && v.vec_ok && (v.ld % 8) == 0 && v.R < (1L << 31) && v.Cc < (1L << 31); };
  auto patch = [&](const GemmView& v) {
    return v.conv == 2 && v.vec_ok && v.KH == v.stride && v.KW == v.stride && v.pad == 0 && v.dil == 1 && v.in_dil <= 1 &&
           v.H == v.OH * v.stride && v.W == v.OW * v.stride && ((long)v.KW * v.C) % 64 == 0 && v.R < (1L << 31) &&
           (long)v.stride * v.W * v.C * 2 < (1L << 31) && (long)v.OH * v.OW < (1L << 31) && p.K == (long)v.KH * v.KW * v.C;
  };
  return p.dtype == CMDA_BF16 && !p.a_kstrided && !p.b_kstrided && (plain(p.A) || patch(p.A)) && plain(p.B) && (p.K % 64) == 0 && p.K >= 64 &&
         (p.N % 4) == 0 && p.c_vec_ok && p.batch == 1 && p.batch2 <= 1 && p.splits <= 1 && !p.atomic && !p.colsum && !p.colstats &&
         p.c_perm_ci == 0 && p.c_patch_ow == 0 && (long)p.M < (1L << 31);
}

//   id  tile      consumers (wave tile)   producers  stages
int cmda_gemm_lean2_(const cmda_gemm_params_t& p, int cfg, void* stream) {
  switch (cfg) {
    case 0:  return launch_lean2<1, 2, 4, 2, 0, 2>(p, stream);   //  64 x 64   8 x (16 x 32)   --   2   (the first lean kernel's shape)
    case 1:  return launch_lean2<1, 2, 4, 2, 0, 3>(p, stream);   //  64 x 64   8 x (16 x 32)   --   3
    case 2:  return launch_lean2<2, 2, 2, 2, 4, 3>(p, stream);   //  64 x 64   4 x (32 x 32)    4   3
    case 3:  return launch_lean2<1, 5, 4, 1, 4, 3>(p, stream);   //  64 x 80   4 x (16 x 80)    4   3
    case 4:  return launch_lean2<2, 1, 2, 5, 0, 3>(p, stream);   //  64 x 80  10 x (32 x 16)   --   3
    case 5:  return launch_lean2<1, 5, 8, 1, 0, 3>(p, stream);   // 128 x 80   8 x (16 x 80)   --   3
    case 6:  return launch_lean2<2, 5, 4, 1, 4, 3>(p, stream);   // 128 x 80   4 x (32 x 80)    4   3
    case 7:  return launch_lean2<2, 5, 4, 2, 0, 2>(p, stream);   // 128 x 160  8 x (32 x 80)   --   2
    case 8:  return launch_lean2<2, 5, 4, 2, 0, 3>(p, stream);   // 128 x 160  8 x (32 x 80)   --   3
    case 9:  return launch_lean2<2, 5, 4, 2, 4, 3>(p, stream);   // 128 x 160  8 x (32 x 80)    4   3
    case 10: return launch_lean2<4, 5, 4, 2, 0, 2>(p, stream);   // 256 x 160  8 x (64 x 80)   --   2
    case 11: return launch_lean2<4, 5, 4, 2, 0, 3>(p, stream);   // 256 x 160  8 x (64 x 80)   --   3   (with producers: 12 waves leave 168 registers per lane, the 64 x 80 wave tile needs more)
    case 12: return launch_lean2<2, 2, 4, 2, 0, 3>(p, stream);   // 128 x 64   8 x (32 x 32)   --   3
    case 13: return launch_lean2<2, 4, 4, 2, 0, 3>(p, stream);   // 128 x 128  8 x (32 x 64)   --   3
    case 14: return launch_lean2<2, 4, 4, 2, 4, 3>(p, stream);   // 128 x 128  8 x (32 x 64)    4   3
    case 15: return launch_lean2<1, 5, 4, 1, 0, 3>(p, stream);   //  64 x 80   4 x (16 x 80)   --   3
    case 16: return launch_lean2<1, 2, 2, 2, 4, 3>(p, stream);   //  32 x 64   4 x (16 x 32)    4   3
    case 17: return launch_lean2<1, 5, 2, 1, 2, 3>(p, stream);   //  32 x 80   2 x (16 x 80)    2   3
    default: return CMDA_ERR_UNSUPPORTED;
  }
}
