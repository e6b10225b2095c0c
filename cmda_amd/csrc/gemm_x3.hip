// gemm_x3.hip -- the parity mode's GEMM on SPLIT-bf16 operands ("bf16 x 3"): fp32 storage in HBM, every operand element split in
// the staging pass into hi = bf16(x) and lo = bf16(x - hi), and each 16x16x32 k-step computed as three bf16 MFMAs with fp32
// accumulate:   a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi      (the dropped a_lo*b_lo term and the rounding of the lo parts are
// ~2^-17 of |a||b| each: ~16 mantissa bits per product, ~1e-5 relative on a contraction).
// Replaces, for cmda_gemm_params_t.dtype == CMDA_F32X3, the exact-fp32 kernel (v_mfma_f32_16x16x4_f32 at 1/16 of the bf16 MFMA
// rate, gemm_kernels.h gemm_kernel<float>): 3/16 of its matrix-pipe time for the same operand bytes -- the mode that meets the
// north star's 1e-3 logit tolerance without paying the fp32 matrix rate.  Same operand views, same fused epilogue,
// same tile / split-K heuristics (gemm.hip launch_dtype<float>); reference call sites as gemm_kernels.h.
#include "gemm_kernels.h"

namespace {

// fp32 chunk (4 values) -> 4 hi + 4 lo bf16
static __device__ __forceinline__ void split4(const uint4& c, u16x4& hi, u16x4& lo) {
  const float f[4] = {__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z), __uint_as_float(c.w)};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bf16_t h = f2bf(f[e]);
    hi[e] = h;
    lo[e] = f2bf(f[e] - bf2f(h));
  }
}

template <int TM, int TN, bool AKS, bool BKS>
__global__ __launch_bounds__(256, 2) void gemm_x3_kernel(GemmParams p) {
  typedef float T;                      // storage type of operands, residual and (unless out_f32 is irrelevant) the output
  constexpr int CH = 4, BM = 32 * TM, BN = 32 * TN, BK = 32;
  constexpr int ROWS_A = AKS ? BK : BM, COLS_A = AKS ? BM : BK, PITCH_A = COLS_A + 8;   // bf16 LDS tiles (hi and lo)
  constexpr int ROWS_B = BKS ? BK : BN, COLS_B = BKS ? BN : BK, PITCH_B = COLS_B + 8;
  constexpr int NCH_A = ROWS_A * (COLS_A / CH) / 256, NCH_B = ROWS_B * (COLS_B / CH) / 256;
  static_assert(NCH_A >= 1 && NCH_B >= 1, "tile too small for 256 threads");
  constexpr int SZ_A = ROWS_A * PITCH_A, SZ_B = ROWS_B * PITCH_B;     // bf16 elements per (stage, part)
  constexpr int PITCH_C = BN + 4;
  constexpr size_t STAGE_BYTES = (size_t)2 * 2 * (SZ_A + SZ_B) * sizeof(bf16_t);   // 2 stages x (hi, lo)
  constexpr size_t EPI_BYTES = (size_t)BM * PITCH_C * sizeof(float);
  constexpr size_t LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  // layout: [stage][A hi | A lo | B hi | B lo]
  constexpr int STAGE_ELEMS = 2 * (SZ_A + SZ_B);
  bf16_t* const sbase = reinterpret_cast<bf16_t*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, g = lane >> 4, l15 = lane & 15;
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

  constexpr int PF = (TM * TN <= 8) ? 2 : 1;
  Stager<T, NCH_A, AKS, BM, BK, PF> stA;
  Stager<T, NCH_B, BKS, BN, BK, PF> stB;
  stA.init(p.A, tid, m0, kt0);
  stB.init(p.B, tid, n0, kt0);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging: the Stager's chunk (row0 + i*RSTEP, cc) of 4 fp32 -> 8 bytes of the hi tile + 8 bytes of the lo tile
  auto put = [&](auto& st, auto S, bf16_t* hi_t, bf16_t* lo_t, auto PITCHc) {
    constexpr int PITCH = decltype(PITCHc)::value;
    constexpr int SS = decltype(S)::value;
    typedef typename std::remove_reference<decltype(st)>::type ST;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(st.reg[0]) / sizeof(uint4)); ++i) {
      u16x4 h, l;
      split4(st.reg[SS][i], h, l);
      const int off = (st.row0 + i * ST::RSTEP) * PITCH + st.cc * CH;
      *reinterpret_cast<u16x4*>(&hi_t[off]) = h;
      *reinterpret_cast<u16x4*>(&lo_t[off]) = l;
    }
  };
  auto store = [&](auto S, int stage) {
    bf16_t* s = sbase + stage * STAGE_ELEMS;
    put(stA, S, s, s + SZ_A, std::integral_constant<int, PITCH_A>());
    put(stB, S, s + 2 * SZ_A, s + 2 * SZ_A + SZ_B, std::integral_constant<int, PITCH_B>());
  };
  auto frag = [&](const bf16_t* t, bool ks, int pitch, int r0) -> u16x8 {   // 8 consecutive k of row/column r0 + l15
    if (!ks) return *reinterpret_cast<const u16x8*>(&t[(r0 + l15) * pitch + 8 * g]);
    const int q = l15 >> 2, pp = l15 & 3;
    const u16x4 lo = lds_read_tr16(&t[(8 * g + q) * pitch + r0 + 4 * pp]);
    const u16x4 hi = lds_read_tr16(&t[(8 * g + 4 + q) * pitch + r0 + 4 * pp]);
    return u16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };
  auto compute = [&](int stage) {
    const bf16_t* s = sbase + stage * STAGE_ELEMS;
    const bf16_t *aH = s, *aL = s + SZ_A, *bH = s + 2 * SZ_A, *bL = s + 2 * SZ_A + SZ_B;
    u16x8 fah[TM], fal[TM], fbh[TN], fbl[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mr = wm * 16 * TM + i * 16;
      fah[i] = frag(aH, AKS, PITCH_A, mr);
      fal[i] = frag(aL, AKS, PITCH_A, mr);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nr = wn * 16 * TN + j * 16;
      fbh[j] = frag(bH, BKS, PITCH_B, nr);
      fbl[j] = frag(bL, BKS, PITCH_B, nr);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {   // small terms first, then the leading product
        acc[i][j] = mfma_bf16_16x16x32(fal[i], fbh[j], acc[i][j]);
        acc[i][j] = mfma_bf16_16x16x32(fah[i], fbl[j], acc[i][j]);
        acc[i][j] = mfma_bf16_16x16x32(fah[i], fbh[j], acc[i][j]);
      }
  };
  typedef std::integral_constant<int, 0> S0;
  typedef std::integral_constant<int, 1> S1;
  if constexpr (PF == 2) {   // two k-tiles ahead in registers, ONE barrier per k-tile (gemm_kernel's pipeline)
    if (kt0 < kt1) {
      stA.template load<0>(p.A, baseA, m0, kt0);
      stB.template load<0>(p.B, baseB, n0, kt0);
      store(S0(), 0);
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
      compute(0);
      if (kt + 1 < kt1) store(S1(), 1);
      __syncthreads();
      if (kt + 1 >= kt1) break;
      if (kt + 3 < kt1) {
        stA.template load<1>(p.A, baseA, m0, kt + 3);
        stB.template load<1>(p.B, baseB, n0, kt + 3);
      }
      compute(1);
      if (kt + 2 < kt1) store(S0(), 0);
      __syncthreads();
    }
  } else {
    if (kt0 < kt1) {
      stA.template load<0>(p.A, baseA, m0, kt0);
      stB.template load<0>(p.B, baseB, n0, kt0);
      store(S0(), 0);
    }
    __syncthreads();
    int cur = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
      if (kt + 1 < kt1) {
        stA.template load<0>(p.A, baseA, m0, kt + 1);
        stB.template load<0>(p.B, baseB, n0, kt + 1);
      }
      compute(cur);
      if (kt + 1 < kt1) store(S0(), cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }

  if (kt0 >= kt1 && p.splits > 1) return;
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

template <int TM, int TN>
int launch_x3(const GemmParams& p, void* stream) {
  constexpr int BM = 32 * TM, BN = 32 * TN;
  const long tiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  if (tiles > 0x7fffffffL || (long)p.batch * p.batch2 * p.splits > 65535) return CMDA_ERR_SHAPE;
  dim3 grid((unsigned)tiles, 1, (unsigned)(p.batch * p.batch2 * p.splits));
  const bool aks = p.a_kstrided != 0, bks = p.b_kstrided != 0;
  if (!aks && !bks) CMDA_LAUNCH((gemm_x3_kernel<TM, TN, false, false>), grid, dim3(256), 0, stream, p);
  else if (!aks && bks) CMDA_LAUNCH((gemm_x3_kernel<TM, TN, false, true>), grid, dim3(256), 0, stream, p);
  else if (aks && bks) CMDA_LAUNCH((gemm_x3_kernel<TM, TN, true, true>), grid, dim3(256), 0, stream, p);
  else CMDA_LAUNCH((gemm_x3_kernel<TM, TN, true, false>), grid, dim3(256), 0, stream, p);
  CMDA_CHECK_LAUNCH();
}

}  // namespace

int cmda_gemm_x3_(const cmda_gemm_params_t& p, int tile, void* stream) {
  const int t = (tile == 3) ? 0 : (tile == 4 ? 2 : tile);
  return t == 0 ? launch_x3<4, 4>(p, stream) : t == 1 ? launch_x3<4, 2>(p, stream) : launch_x3<2, 2>(p, stream);
}
