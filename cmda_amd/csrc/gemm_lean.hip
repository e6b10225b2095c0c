// gemm_lean.hip -- the LEAN instance of the LDS-DMA GEMM for the launches that dominate the UDA step's count: the encoders' Linear
// layers and their data gradients at 2 + 2 samples per GPU (q / kv / proj / fc1 / fc2 of mix_transformer.py:31-44,62-66,80-102, ~1.8 k
// launches per step of 0.2 - 7 GFLOP each).  Same tiles, same LDS image, same k-loop and the same fused epilogue as
// gemm_glds_kernel (gemm_kernels.h) -- what differs is everything AROUND the k-loop, which at these sizes is most of the kernel:
// the phase stamps of the general kernel on 4096 x 320 x 320 (tools/gemm_phase.py, gpurun r04h) read
//     setup 1.28 us | first stage 0.16 | k-loop 1.60 | stage-C 0.36 | store 1.48      (block total 4.9 us)
// -- the 300-byte parameter block arrives in six dependent scalar-load round trips spread over the branches of the general prologue
// (operand-view modes, batch / split-K / patch / im2col arithmetic, the tile-group walk: ~1900 instructions ahead of the first DMA),
// and the epilogue starts its bias / residual loads only when the accumulators are final.  Here the host digests the problem into a
// 128-byte block that is read in ONE round trip, the prologue is straight-line pointer arithmetic, and the epilogue's bias and
// residual loads are issued before the k-loop's last barrier.
// Eligibility (host, cmda_gemm -> launch_dtype): bf16, plain operands (no im2col / patch view), A K-contiguous, B K-contiguous or
// K-strided, K % 64 == 0, no batch, no split-K, no atomic / patch-store / column-sum output; tiles 64 x 64 and 128 x 64.
#include "gemm_kernels.h"

namespace {

struct LeanParams {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const float* bias;
  const void* res;
  const float* rowscale;
  long lda, ldb, ldc, ldres;
  int M, N, nkt, tiles_n;
  int ntile, rows_per_scale, act, flags;   // flags: 1 out_f32, 2 res_f32, 4 c_vec_ok, 8 patch store (p_ow / p_row / p_col = OW / KH / KW*Ci)
  float alpha, beta;
  // patch view of A (kernel == stride convolution, cmda_view_t.conv == 2): row m = (b, oh, ow) starts at b * p_img + oh * p_row + ow * p_col
  // elements; its K = KH * KW * C elements are KH segments of p_seg k-tiles (KW * C contiguous elements) p_jump bytes apart
  long p_img;
  int p_ohw, p_ow, p_row, p_col, p_seg, p_jump;
};

// NSV = 0: two LDS stages (most resident blocks); NSV = 4: the latency configuration for grids that are resident at once and run
// >= 12 k-tiles per block (fc2 / its data gradient at K = 1280 ... 2048: three k-tiles in flight) -- launch_glds's rule
// PATCH: A is the patch view of a kernel == stride convolution (the spatial-reduction convolutions of the MiT attention,
// mix_transformer.py:73-75,86-90): the running pointer jumps to the next input row at the end of every KW * C segment
// -DCMDA_LEAN_TIMING (tuning build `make leantiming`, tools/dbg/lean_phase.py): lane 0 of wave 0 of workgroup 0 stamps s_memtime at entry (0),
// after the prologue's DMA issue (1), at the top of k-tile kt (4 + kt, kt < 40), at the end of the k-loop (50), when the accumulators went
// through LDS (51), at the end of the epilogue (52)
#ifdef CMDA_LEAN_TIMING
__device__ unsigned long long g_lean_stamps[64];
#define LEAN_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.z == 0) g_lean_stamps[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LEAN_STAMP(i) do { } while (0)
#endif

template <int TM, int TN, bool BKS, int NSV = 0, int NW = 4, bool PATCH = false>
__global__ __launch_bounds__(64 * NW, (GldsCfg<TM, TN, NW, NSV>::MIN_WAVES)) void gemm_lean_kernel(LeanParams q) {
  LEAN_STAMP(0);
  typedef bf16_t T;
  typedef GldsCfg<TM, TN, NW, NSV> Cfg;
  constexpr int NT = 64 * NW;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = 64, NS = Cfg::NS;
  constexpr int SZ_A = Cfg::SZ_A, SZ_B = Cfg::SZ_B, PITCH_C = Cfg::PITCH_C;
  __shared__ __attribute__((aligned(1024))) char smem[Cfg::LDS_BYTES];
  T* const sAbase = reinterpret_cast<T*>(smem);
  T* const sBbase = sAbase + NS * SZ_A;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, g = lane >> 4, l15 = lane & 15;
  // every field of the parameter block is consumed in this straight-line prologue (one scalar-load round trip)
  const int M = q.M, N = q.N, nkt = q.nkt, tiles_n = q.tiles_n, ntile = q.ntile;
  const long lda = q.lda, ldb = q.ldb;
  int bt = blockIdx.x;
  {   // XCD-contiguous tile ranges (blocks b and b + 8 share an L2): n-tiles of one m-panel first
    const int qq = ntile >> 3, rr = ntile & 7, xcd = bt & 7, loc = bt >> 3;
    bt = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + loc;
  }
  const int mt = (int)((unsigned)bt / (unsigned)tiles_n), nt = bt - mt * tiles_n;
  const long m0 = (long)mt * BM, n0 = (long)nt * BN;

  // DMA sources: running pointers (gemm_kernels.h DmaSrc mode 1), the LDS image of gemm_glds_body: line r, slot = chunk ^ (r & 7)
  constexpr int JA = BM * BK / (512 * NW), JB = BN * BK / (512 * NW);
  static_assert(JA >= 1 && JB >= 1, "tile too small for the wave count");
  constexpr int CPL_B = BKS ? BN / 8 : 8, LPI_B = 64 / CPL_B;
  const char* curA[JA];
  const char* curB[JB];
  int stepA[JA], stepB[JB];
  const char* zero = reinterpret_cast<const char*>(g_zero16);
#pragma unroll
  for (int j = 0; j < JA; ++j) {
    const int ln = (wid * JA + j) * 8 + (lane >> 3), chunk = (lane & 7) ^ (ln & 7);
    if constexpr (!PATCH) {
      const long r = m0 + ln;
      const bool ok = r < M;
      curA[j] = ok ? reinterpret_cast<const char*>(q.A + r * lda + chunk * 8) : zero;
      stepA[j] = ok ? BK * 2 : 0;
    } else {   // rows past M read row M - 1 (the segment jump is uniform; the epilogue never stores them)
      const unsigned r = (unsigned)min(m0 + ln, (long)M - 1);
      const unsigned b = r / (unsigned)q.p_ohw, rem = r - b * (unsigned)q.p_ohw, oh = rem / (unsigned)q.p_ow, ow = rem - oh * (unsigned)q.p_ow;
      curA[j] = reinterpret_cast<const char*>(q.A + (long)b * q.p_img + (long)oh * q.p_row + (long)ow * q.p_col + chunk * 8);
      stepA[j] = BK * 2;
    }
  }
  int seg_left = PATCH ? q.p_seg : 0;
#pragma unroll
  for (int j = 0; j < JB; ++j) {
    const int ln = (wid * JB + j) * LPI_B + lane / CPL_B, chunk = (lane % CPL_B) ^ (ln & 7);
    if constexpr (!BKS) {
      const long r = n0 + ln;
      const bool ok = r < N;
      curB[j] = ok ? reinterpret_cast<const char*>(q.B + r * ldb + chunk * 8) : zero;
      stepB[j] = ok ? BK * 2 : 0;
    } else {
      const long c = n0 + chunk * 8;
      const bool ok = c + 8 <= N;
      curB[j] = ok ? reinterpret_cast<const char*>(q.B + (long)ln * ldb + c) : zero;
      stepB[j] = ok ? (int)((long)BK * ldb * 2) : 0;
    }
  }
  auto issue = [&](int stage) {
    char* la = reinterpret_cast<char*>(sAbase + stage * SZ_A) + wid * JA * 1024;
    char* lb = reinterpret_cast<char*>(sBbase + stage * SZ_B) + wid * JB * 1024;
#pragma unroll
    for (int j = 0; j < JA; ++j) {
      glds16(curA[j], la + j * 1024);
      curA[j] += stepA[j];
    }
    if constexpr (PATCH) {
      if (--seg_left == 0) {
        seg_left = q.p_seg;
#pragma unroll
        for (int j = 0; j < JA; ++j) curA[j] += q.p_jump;
      }
    }
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      glds16(curB[j], lb + j * 1024);
      curB[j] += stepB[j];
    }
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nkt) issue(s);
  LEAN_STAMP(1);

  // epilogue operands requested NOW: the thread's bias quad (and, below, its residual rows before the last barrier)
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

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int LPT = JA + JB;           // DMA instructions per wave per k-tile
  int st = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt < 40) LEAN_STAMP(4 + kt);
    // tile kt has landed (NS - 2 younger tiles may stay in flight; the tail drains); every wave is done reading stage st - 1
    if (kt + NS - 2 < nkt) pipe_barrier<(NS - 2) * LPT>();
    else pipe_barrier<0>();
    {
      int sn = st + NS - 1;
      if (sn >= NS) sn -= NS;
      if (kt + NS - 1 < nkt) issue(sn);
    }
    const T* sA = sAbase + st * SZ_A;
    const T* sB = sBbase + st * SZ_B;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      u16x8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * 16 * TM + i * 16 + l15;
        fa[i] = *reinterpret_cast<const u16x8*>(&sA[row * BK + (((kk * 4 + g) ^ (row & 7)) << 3)]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int nr = wn * 16 * TN + j * 16;
        if constexpr (!BKS) {
          const int row = nr + l15;
          fb[j] = *reinterpret_cast<const u16x8*>(&sB[row * BK + (((kk * 4 + g) ^ (row & 7)) << 3)]);
        } else {
          const int qd = l15 >> 2, pp = l15 & 3;
          const int k0 = kk * 32 + 8 * g + qd, k1 = k0 + 4;
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
  // residual rows of this thread: requested before the accumulators go through LDS (their latency hides behind the staging)
  const bool has_res = q.res != nullptr, res32 = (q.flags & 2) != 0, f32o = (q.flags & 1) != 0;
  float rv[NIT][4];
LEAN_STAMP(50);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    rv[it][0] = rv[it][1] = rv[it][2] = rv[it][3] = 0.f;
    const long m = m0 + er0 + it * RSTEP;
    if (has_res && ecol && m < M) {
      const long ri = m * q.ldres + en;
      if (full) {
        if (res32) ld4(reinterpret_cast<const float*>(q.res) + ri, rv[it]);
        else ld4(reinterpret_cast<const T*>(q.res) + ri, rv[it]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (en + e < N) rv[it][e] = res32 ? reinterpret_cast<const float*>(q.res)[ri + e] : ldf(reinterpret_cast<const T*>(q.res) + ri + e);
      }
    }
  }
  __syncthreads();
  float* sC = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        sC[(wm * 16 * TM + i * 16 + 4 * g + r) * PITCH_C + wn * 16 * TN + j * 16 + l15] = acc[i][j][r];
  __syncthreads();
  LEAN_STAMP(51);
  if (!ecol) return;
  const float alpha = q.alpha, beta = q.beta;
  const bool has_beta = beta != 0.f, has_rs = q.rowscale != nullptr;
  const int act = q.act;
  // patch store (flags bit 3; plain A only, so the patch-view fields are free: p_ow = OW, p_row = KH, p_col = KW * Ci): the data
  // gradient of a kernel == stride convolution goes straight into the NHWC input gradient (mix_transformer.py:70-75 under autograd)
  const bool cpatch = !PATCH && (q.flags & 8) != 0;
  // the common epilogue -- whole column quads, plain store, bias (+ residual), no activation / row scale / beta: the Linear layers
  // and data gradients of the encoders -- as a loop without the general one's per-row flag tests.  Phase stamps (`make leantiming`):
  // the general loop ran ~1000 cycles per row iteration, 21 % of a 4096 x 320 x 320 workgroup, instruction-bound on its uniform branches
  if (full && !cpatch && !has_rs && !has_beta && act == 0) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = er0 + it * RSTEP;
      const long m = m0 + row;
      if (m >= M) break;
      const float4 t = *reinterpret_cast<const float4*>(&sC[row * PITCH_C + q4]);
      float v[4] = {alpha * t.x + bv[0] + rv[it][0], alpha * t.y + bv[1] + rv[it][1], alpha * t.z + bv[2] + rv[it][2], alpha * t.w + bv[3] + rv[it][3]};
      const long ci = m * q.ldc + en;
      if (f32o) st4(reinterpret_cast<float*>(q.C) + ci, v);
      else st4(reinterpret_cast<T*>(q.C) + ci, v);
    }
    LEAN_STAMP(52);
    return;
  }
  unsigned pkh = 0, prest = 0;
  if (cpatch) {
    pkh = (unsigned)en / (unsigned)q.p_col;
    prest = (unsigned)en - pkh * (unsigned)q.p_col;
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int row = er0 + it * RSTEP;
    const long m = m0 + row;
    if (m >= M) break;
    const float4 t = *reinterpret_cast<const float4*>(&sC[row * PITCH_C + q4]);
    float v[4] = {t.x, t.y, t.z, t.w};
    long ci = m * q.ldc + en;
    if (cpatch) {   // un-patchify (cmda_gemm_params_t.c_patch_*): row (b*OH + oh, ow), column (kh, kw*Ci + ci) -> NHWC
      const unsigned boh = (unsigned)m / (unsigned)q.p_ow, ow = (unsigned)m - boh * (unsigned)q.p_ow;
      ci = ((long)(boh * (unsigned)q.p_row + pkh) * q.p_ow + ow) * q.p_col + prest;
    }
    float ov[4] = {0.f, 0.f, 0.f, 0.f};
    float rs = 1.f;
    if (has_rs) rs = q.rowscale[(unsigned)m / (unsigned)q.rows_per_scale];   // (32-bit: M < 2^31)
    if (has_beta) {
      if (full) {
        if (f32o) ld4(reinterpret_cast<const float*>(q.C) + ci, ov);
        else ld4(reinterpret_cast<const T*>(q.C) + ci, ov);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (en + e < N) ov[e] = f32o ? reinterpret_cast<const float*>(q.C)[ci + e] : ldf(reinterpret_cast<const T*>(q.C) + ci + e);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = alpha * v[e] + bv[e];
      const float a = act == 0 ? x : act == 1 ? epi_act<1>(x) : act == 2 ? epi_act<2>(x) : epi_act<3>(x);
      v[e] = a * rs + rv[it][e] + beta * ov[e];
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
  LEAN_STAMP(52);
}

template <int TM, int TN, int NSV = 0, int NW = 4>
int launch_lean(const GemmParams& p, void* stream) {
  constexpr int BM = 16 * TM * (NW / 2), BN = 32 * TN;
  LeanParams q;
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
  q.flags = (p.out_f32 ? 1 : 0) | (p.res_f32 ? 2 : 0) | (p.c_vec_ok ? 4 : 0);
  q.alpha = p.alpha; q.beta = p.beta;
  const dim3 grid((unsigned)tiles), blk(64 * NW);
  if (p.A.conv == 2) {
    if constexpr (NSV == 0 && NW == 4) return CMDA_ERR_UNSUPPORTED;   // (the four-wave 2-stage tuning variants: plain operands only)
    else {
      const GemmView& v = p.A;
      q.p_img = (long)v.H * v.W * v.C;
      q.p_ohw = v.OH * v.OW; q.p_ow = v.OW;
      q.p_row = v.stride * v.W * v.C; q.p_col = v.stride * v.C;
      q.p_seg = v.KW * v.C / 64;
      q.p_jump = (v.W - v.KW) * v.C * 2;
      CMDA_LAUNCH((gemm_lean_kernel<TM, TN, false, NSV, NW, true>), grid, blk, 0, stream, q);
      CMDA_CHECK_LAUNCH();
    }
  }
  if (p.c_patch_ow > 0) {
    q.flags |= 8;
    q.p_ow = p.c_patch_ow; q.p_row = p.c_patch_kh; q.p_col = p.c_patch_kwci;
  }
  if (p.b_kstrided) CMDA_LAUNCH((gemm_lean_kernel<TM, TN, true, NSV, NW>), grid, blk, 0, stream, q);
  else CMDA_LAUNCH((gemm_lean_kernel<TM, TN, false, NSV, NW>), grid, blk, 0, stream, q);
  CMDA_CHECK_LAUNCH();
}

}  // namespace

// HOST: does the lean kernel take this problem?  (tile: launch_dtype's choice, 1 = 128 x 64, 2 = 64 x 64)
bool cmda_gemm_lean_ok_(const cmda_gemm_params_t& p, int tile) {
  auto plain = [](const GemmView& v) { return v.conv == 0 && v.vec_ok && (v.ld % 8) == 0 && v.R < (1L << 31) && v.Cc < (1L << 31); };
  // patch view of a kernel == stride convolution as A (K-contiguous B): KW * C a multiple of the k-tile, 32-bit row arithmetic
  auto patch = [&](const GemmView& v) {
    return v.conv == 2 && v.vec_ok && !p.b_kstrided && v.KH == v.stride && v.KW == v.stride && v.pad == 0 && v.dil == 1 && v.in_dil <= 1 &&
           v.H == v.OH * v.stride && v.W == v.OW * v.stride && ((long)v.KW * v.C) % 64 == 0 && v.R < (1L << 31) &&
           (long)v.stride * v.W * v.C * 2 < (1L << 31) && (long)v.OH * v.OW < (1L << 31) && p.K == (long)v.KH * v.KW * v.C &&
           !(p.tile_hint > 0 && (p.tile_hint & 16384));
  };
  return (tile == 1 || tile == 2) && p.dtype == CMDA_BF16 && !p.a_kstrided && (plain(p.A) || patch(p.A)) && plain(p.B) && (p.K % 64) == 0 && p.K >= 64 &&
         p.batch == 1 && p.batch2 <= 1 && p.splits <= 1 && !p.atomic && !p.colsum && p.c_perm_ci == 0 &&
         (p.c_patch_ow == 0 || (plain(p.A) && !p.res && (long)p.M * p.N < (1L << 31) && p.c_patch_kwci > 0)) &&
         (!p.b_kstrided || 64L * p.B.ld * 2 < (1L << 31)) && !(p.tile_hint > 0 && (p.tile_hint & 8192));   // (tile_hint bit 13: general kernel, tuning A/B)
}

int cmda_gemm_lean_(const cmda_gemm_params_t& p, int tile, int four_stage, void* stream) {
  // The 2-stage configurations run on EIGHT waves (16 x 32 / 32 x 32 of the tile per wave): a lone workgroup's k-tile is bound by how
  // fast its waves can issue LDS-DMA pieces (0.3 us per 16 KiB with four waves), and twice the waves issue half the pieces each --
  // 2048 x 320 x 320 5.05 -> 4.31 us, 8192 x 1280 x 320 18.6 -> 16.6, the step 60.2 -> 59.0 ms (gpurun r04u; the 4-stage
  // configurations measured the same on 4 and 8 waves and stay on 4).  tile_hint bit 14: four waves (tuning A/B).
  const bool w4 = p.tile_hint > 0 && (p.tile_hint & 16384);
  if (!four_stage && !w4) return tile == 1 ? launch_lean<2, 2, 0, 8>(p, stream) : launch_lean<1, 2, 0, 8>(p, stream);
  if (four_stage) return tile == 1 ? launch_lean<4, 2, 4>(p, stream) : launch_lean<2, 2, 4>(p, stream);
  return tile == 1 ? launch_lean<4, 2>(p, stream) : launch_lean<2, 2>(p, stream);
}

#ifdef CMDA_LEAN_TIMING
extern "C" int cmda_debug_lean_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_lean_stamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -3;
}
#endif
