// gemm_x3_lean.hip -- the LEAN instance of the split-bf16 ("bf16 x 3") GEMM: the encoders' Linear layers and their data gradients in
// the tolerance-meeting mode (fp32 storage, CMDA_F32X3; q / kv / proj / fc1 / fc2 of mix_transformer.py:31-44,62-66,80-102).
//
// Why: profiles/r04_x3_eager_kernel_stats.csv -- 2 265 launches per step of gemm_x3_kernel<2, 2, ...> at 37 - 46 us each (96 ms of
// the step's 209) for problems the bf16 lean kernel runs in 8 - 10 us.  That kernel is register-staged (global -> VGPR -> split ->
// LDS) on 32-deep k-tiles with four waves: ten exposed global-load latencies for K = 320.  Here the fp32 tiles come in through the
// LDS-DMA (32-deep k-tiles, two tiles in flight while a third is consumed), are split ONCE per tile by all 512 threads
// (x -> hi = bf16(x), lo = bf16(x - hi): one pass over the tile instead of once per fragment and wave) into bf16 hi / lo tiles laid
// out exactly like the bf16 kernels' LDS images, and the k-loop is gemm_lean_kernel's with three MFMAs per fragment pair
// (a_lo b_hi + a_hi b_lo + a_hi b_hi, fp32 accumulate: gemm_x3.hip).  The split of tile kt + 1 and the MFMAs of tile kt sit in the
// same barrier interval (double-buffered hi / lo tiles), so the VALU pass hides under the matrix pipe.
// Eligibility (host): CMDA_F32X3, plain operands, K % 32 == 0, no patch-store output (batch x batch2 entries = grid.z: the unfused attention GEMMs); A K-contiguous with B K-contiguous or
// K-strided (Linear forward / data gradient: no split-K, atomics or column sums), or BOTH K-strided with atomic output (weight
// gradient: split-K chosen here, bias gradient fused); 64 x 64 tiles on eight waves, chosen where the general kernel would run its 64 x 64 or
// 128 x 64 tile (the small grids of the encoders).
#include "gemm_kernels.h"

namespace {

// -DCMDA_X3_TIMING (tuning build `make x3timing`, tools/dbg/x3_phase.py): lane 0 of wave 0 of workgroup 0 stamps s_memtime at entry (0),
// after the prologue's DMA issue (1), when tile 0 has landed (2), after the first split (3), at the top of k-tile kt (4 + kt, kt < 40),
// at the end of the k-loop (50), after the accumulators went through LDS (51), at the end of the epilogue (52)
#ifdef CMDA_X3_TIMING
__device__ unsigned long long g_x3_stamps[64];
#define X3_STAMP(i) do { if (threadIdx.x == 0 && bt_in == 0 && zidx == 0) g_x3_stamps[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define X3_STAMP(i) do { } while (0)
#endif

struct X3LeanParams {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* res;
  const float* rowscale;
  long lda, ldb, ldc, ldres;
  int M, N, nkt, tiles_n;
  int ntile, rows_per_scale, act, flags;   // flags: 4 c_vec_ok
  float alpha, beta;
  float* colsum;                           // weight-gradient form: bias gradient [M] += sum over k of A(k, m) (workgroups of n-tile 0)
  int kt_per;                              // weight-gradient form: k-tiles per split
  int splits, batch2;                      // grid.z = batch * batch2 * splits (the attention GEMMs of the unfused fp32 path: batch x heads)
  long a_bs, a_b2s, b_bs, b_b2s, c_bs, c_b2s, r_bs, r_b2s;   // element strides of the outer / inner batch index
  // patch view of A (kernel == stride convolution, cmda_view_t.conv == 2: the spatial-reduction convolutions): row m = (b, oh, ow) starts
  // at b * p_img + oh * p_row + ow * p_col elements; its K = KH * KW * C elements are KH segments of p_seg k-tiles (KW * C contiguous
  // elements) p_jump bytes apart (gemm_lean.hip); p_seg == 0: plain operand
  long p_img;
  int p_ohw, p_ow, p_row, p_col, p_seg, p_jump;
  // forward form with atomic != 0: split-K over grid.z (kt_per k-tiles per split), fp32 atomics into C (the spatial-reduction
  // convolutions of small token counts: bias rows pre-filled by cmda_rows_fill)
  int atomic;
  // patch-store epilogue (cmda_gemm_params_t.c_patch_*: the data gradient of a kernel == stride convolution stored straight into NHWC)
  int c_patch_ow, c_patch_kh, c_patch_kwci;
  // weight-gradient form with a PATCH view as B (dW of a kernel == stride convolution: B(k = token, n = (kh, kw, ci))): token
  // (b, oh, ow) starts at b * pb_img + oh * pb_row + ow * pb_col elements; column n = kh * pb_kwc + rest lies at kh * pb_line + rest
  long pb_img;
  int pb_ohw, pb_ow, pb_row, pb_col, pb_kwc, pb_line;   // pb_kwc == 0: plain B
};

// 64 x 64 tile on eight waves = 4 (rows of 16) x 2 (columns of 32); 32-deep k-tiles: the fp32 values of two tiles wait in registers
// (16 bytes of A and of B per thread and tile), two hi / lo sets of 16 KiB in LDS
// AKS (with BKS): the WEIGHT-GRADIENT form -- both operands K-strided ([tokens][channels] row-major: dW = dy^T x), split-K over the
// tokens (grid.z), fp32 atomic accumulation into C, bias gradient (column sums of dy) fused
// bt_in: output tile of the problem (hardware order; remapped to XCD-contiguous ranges when `remap`), zidx: (batch entry, split)
template <bool AKS, bool BKS>
static __device__ __forceinline__ void x3_lean_body(X3LeanParams& q, int bt_in, int zidx, bool remap) {
  static_assert(!AKS || BKS, "A K-strided: weight-gradient form only");
  X3_STAMP(0);
  constexpr int NW = 8, NT = 512, BM = 64, BN = 64, BK = 32, TN = 2;
  constexpr int H_A = BM * BK, H_B = BN * BK;                 // bf16 elements of a hi (or lo) tile
  constexpr int PITCH_C = BN + 4;
  constexpr size_t HL_BYTES = (size_t)2 * 2 * (H_A + H_B) * 2;   // two (hi, lo) tile sets
  static_assert(HL_BYTES >= (size_t)BM * PITCH_C * 4 && HL_BYTES >= 32 * 64 * 4, "the epilogue's tile / the bias-gradient fold alias the sets");
  __shared__ __attribute__((aligned(1024))) char smem[HL_BYTES];
  bf16_t* const sH = reinterpret_cast<bf16_t*>(smem);         // [set][A hi | A lo | B hi | B lo]
  constexpr int SET = 2 * (H_A + H_B);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, g = lane >> 4, l15 = lane & 15;
  const int M = q.M, N = q.N, tiles_n = q.tiles_n, ntile = q.ntile;
  // flags bit 4: batched (grid.z carries batch entries), bit 5: K split -- the plain Linear / data-gradient launch has neither and
  // skips their divisions and the cold half of the parameter block (a scalar-load round trip of the prologue)
  int bz = 0, ksplit = 0;
  if (q.flags & 48) {
    bz = zidx / q.splits;
    ksplit = zidx - bz * q.splits;
  }
  const int kt0 = ksplit * q.kt_per;                   // (kt_per = all k-tiles and one split unless the launcher split K)
  const int nkt = min(q.nkt - kt0, q.kt_per);          // k-tiles of THIS workgroup
  if (q.flags & 16) {   // batch entry: operand / output bases move, everything else is per problem
    const int b1 = bz / q.batch2, b2 = bz - b1 * q.batch2;
    q.A += b1 * q.a_bs + b2 * q.a_b2s;
    q.B += b1 * q.b_bs + b2 * q.b_b2s;
    q.C += b1 * q.c_bs + b2 * q.c_b2s;
    if (q.res) q.res += b1 * q.r_bs + b2 * q.r_b2s;
  }
  int bt = bt_in;
  if (remap) {   // XCD-contiguous tile ranges (gemm_lean_kernel)
    const int qq = ntile >> 3, rr = ntile & 7, xcd = bt & 7, loc = bt >> 3;
    bt = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + loc;
  }
  const int mt = (int)((unsigned)bt / (unsigned)tiles_n), nt = bt - mt * tiles_n;
  const long m0 = (long)mt * BM, n0 = (long)nt * BN;

  // ---- DMA: an fp32 stage is a LINEAR image of 16-byte positions; thread t owns position t of the A part and position t of the
  //      B part, for the DMA (a wave instruction = 64 positions) and for the split pass alike.
  //      K-contiguous operand: line = row, 8 positions (32 k) per line; K-strided B: line = k, 16 positions (64 columns) per line ----
  const char* curA;
  const char* curB;
  int stepA, stepB;
  const char* zero = reinterpret_cast<const char*>(g_zero16);
  const int lnA = AKS ? tid >> 4 : tid >> 3, chA = AKS ? tid & 15 : tid & 7;
  int seg_left = q.p_seg;
  if constexpr (!AKS) {
    const long r = m0 + lnA;
    const bool ok = r < M;
    if (q.p_seg > 0) {   // patch view: rows past M read row M - 1 (the segment jump stays uniform; the epilogue never stores them)
      const unsigned rr = (unsigned)min(r, (long)M - 1);
      const unsigned pb = rr / (unsigned)q.p_ohw, rem = rr - pb * (unsigned)q.p_ohw, oh = rem / (unsigned)q.p_ow, ow = rem - oh * (unsigned)q.p_ow;
      curA = reinterpret_cast<const char*>(q.A + (long)pb * q.p_img + (long)oh * q.p_row + (long)ow * q.p_col + chA * 4);
      stepA = BK * 4;
      if (kt0 > 0) {   // split-K: this workgroup starts in segment kt0 / p_seg, kt0 % p_seg k-tiles into it
        const int sg = kt0 / q.p_seg, within = kt0 - sg * q.p_seg;
        curA += (long)sg * ((long)q.p_seg * BK * 4 + q.p_jump) + (long)within * BK * 4;
        seg_left = q.p_seg - within;
      }
    } else {
      curA = ok ? reinterpret_cast<const char*>(q.A + r * q.lda + (long)kt0 * BK + chA * 4) : zero;
      stepA = ok ? BK * 4 : 0;
    }
  } else {                  // line = k (token), 64 consecutive output rows m
    const long c = m0 + chA * 4;
    const bool ok = c + 4 <= M;
    curA = ok ? reinterpret_cast<const char*>(q.A + ((long)kt0 * BK + lnA) * q.lda + c) : zero;
    stepA = ok ? (int)((long)BK * q.lda * 4) : 0;
  }
  const int lnB = BKS ? tid >> 4 : tid >> 3, chB = BKS ? tid & 15 : tid & 7;
  if constexpr (!BKS) {
    const long r = n0 + lnB;
    const bool ok = r < N;
    curB = ok ? reinterpret_cast<const char*>(q.B + r * q.ldb + (long)kt0 * BK + chB * 4) : zero;
    stepB = ok ? BK * 4 : 0;
  } else {
    const long c = n0 + chB * 4;
    const bool ok = c + 4 <= N;
    curB = ok ? reinterpret_cast<const char*>(q.B + ((long)kt0 * BK + lnB) * q.ldb + c) : zero;
    stepB = ok ? (int)((long)BK * q.ldb * 4) : 0;
  }
  // patch view as the K-strided B of the weight-gradient form: the thread's token moves 32 rows per k-tile, its column offset
  // (kh segment + rest: a 64-column tile never straddles a segment, KW * C % 64 == 0) is fixed
  unsigned tokB = 0;
  long colB = 0;
  if constexpr (AKS) {
    if (q.pb_kwc > 0) {
      const unsigned c = (unsigned)(n0 + chB * 4), kh = c / (unsigned)q.pb_kwc;
      colB = (long)kh * q.pb_line + (c - kh * (unsigned)q.pb_kwc);
      tokB = (unsigned)(kt0 * BK + lnB);
    }
  }
  auto patch_b = [&]() {
    const unsigned pb = tokB / (unsigned)q.pb_ohw, rem = tokB - pb * (unsigned)q.pb_ohw, oh = rem / (unsigned)q.pb_ow, ow = rem - oh * (unsigned)q.pb_ow;
    tokB += BK;
    return reinterpret_cast<const char*>(q.B + (long)pb * q.pb_img + (long)oh * q.pb_row + (long)ow * q.pb_col + colB);
  };
  // ---- loads: thread t owns 16-byte position t of the A part and of the B part of every k-tile; the fp32 values travel through
  //      REGISTERS (two tiles in flight: one being converted, one on its way), not through LDS stages -- the first version of this
  //      kernel brought them in by LDS-DMA and split them in a second pass: 16 KiB written + 16 KiB read per k-tile on top of the
  //      16 KiB of hi / lo writes and the 48 KiB of fragment reads, and the phase stamps (`make x3timing`) showed the k-tile bound by
  //      exactly that LDS traffic (96 KiB at 128 B per cycle = 768 of its 1200 cycles) ----
  auto fetch = [&](f32x4& ra, f32x4& rb) {
    ra = *reinterpret_cast<const f32x4*>(curA);
    if constexpr (AKS) {
      if (q.pb_kwc > 0) curB = patch_b();
    }
    rb = *reinterpret_cast<const f32x4*>(curB);
    curA += stepA;
    curB += stepB;
    if constexpr (!AKS) {
      if (seg_left > 0 && --seg_left == 0) {   // patch view: the next k-tile starts on the next input row of the patch
        seg_left = q.p_seg;
        curA += q.p_jump;
      }
    }
  };
  // ---- split pass: 16 bytes of fp32 -> 8 bytes of the hi tile + 8 bytes of the lo tile.  bf16 images: K-contiguous [line][32 k]
  //      (64-byte lines of four 16-byte chunks, slot = chunk ^ ((line >> 1) & 3): eight consecutive lines of one k-chunk cover all 32 banks once);
  //      K-strided [32 k][64 columns] (128-byte lines, slot = chunk ^ (k & 7): the bf16 kernels' image) ----
  const int offA = AKS ? lnA * BM + ((((chA >> 1)) ^ (lnA & 7)) << 3) + ((chA & 1) << 2)
                       : lnA * BK + ((((chA >> 1)) ^ ((lnA >> 1) & 3)) << 3) + ((chA & 1) << 2);
  const int offB = BKS ? lnB * BN + ((((chB >> 1)) ^ (lnB & 7)) << 3) + ((chB & 1) << 2)
                       : lnB * BK + ((((chB >> 1)) ^ ((lnB >> 1) & 3)) << 3) + ((chB & 1) << 2);
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  auto convert = [&](const f32x4& ra, const f32x4& rb, int set) {
    bf16_t* ha = sH + set * SET;
    bf16_t* hb = ha + 2 * H_A;
    const float va[4] = {ra[0], ra[1], ra[2], ra[3]}, vb[4] = {rb[0], rb[1], rb[2], rb[3]};
    float hi[4], lo[4];
    if constexpr (AKS) {   // bias gradient: this thread's four columns of dy, every token of its line index
#pragma unroll
      for (int e = 0; e < 4; ++e) csum[e] += va[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hi[e] = bf2f(f2bf(va[e]));
      lo[e] = va[e] - hi[e];
    }
    st4(ha + offA, hi);
    st4(ha + H_A + offA, lo);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hi[e] = bf2f(f2bf(vb[e]));
      lo[e] = vb[e] - hi[e];
    }
    st4(hb + offB, hi);
    st4(hb + H_B + offB, lo);
  };
  f32x4 ra0 = f32x4{0.f, 0.f, 0.f, 0.f}, rb0 = ra0, ra1 = ra0, rb1 = ra0;   // tile kt in slot kt & 1
  if (nkt > 0) fetch(ra0, rb0);
  if (nkt > 1) fetch(ra1, rb1);
  X3_STAMP(1);

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

  // fragments of hi / lo set `set` (requested FIRST in a k-tile: their LDS latency passes under the split pass of the next tile, which
  // the compiler cannot prove disjoint and would otherwise keep strictly in front -- phase stamps of `make x3timing`: 1200 cycles per
  // k-tile of which 730 were this read -> MFMA chain on its own)
  auto load_frags = [&](int set, u16x8& fah, u16x8& fal, u16x8 (&fbh)[TN], u16x8 (&fbl)[TN]) {
    const bf16_t* aH = sH + set * SET;
    const bf16_t* aL = aH + H_A;
    const bf16_t* bH = aH + 2 * H_A;
    const bf16_t* bL = bH + H_B;
    if constexpr (!AKS) {
      const int row = wm * 16 + l15;
      const int off = row * BK + ((g ^ ((row >> 1) & 3)) << 3);
      fah = *reinterpret_cast<const u16x8*>(&aH[off]);
      fal = *reinterpret_cast<const u16x8*>(&aL[off]);
    } else {
      const int mr = wm * 16, qd = l15 >> 2, pp = l15 & 3;
      const int k0 = 8 * g + qd, k1 = k0 + 4;
      const int cidx = (mr >> 3) + (pp >> 1), half = (pp & 1) << 2;
      const int o0 = k0 * BM + ((cidx ^ (k0 & 7)) << 3) + half, o1 = k1 * BM + ((cidx ^ (k1 & 7)) << 3) + half;
      const u16x4 h0 = lds_read_tr16(&aH[o0]), h1 = lds_read_tr16(&aH[o1]);
      const u16x4 l0 = lds_read_tr16(&aL[o0]), l1 = lds_read_tr16(&aL[o1]);
      fah = u16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
      fal = u16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nr = wn * 16 * TN + j * 16;
      if constexpr (!BKS) {
        const int row = nr + l15;
        const int off = row * BK + ((g ^ ((row >> 1) & 3)) << 3);
        fbh[j] = *reinterpret_cast<const u16x8*>(&bH[off]);
        fbl[j] = *reinterpret_cast<const u16x8*>(&bL[off]);
      } else {
        const int qd = l15 >> 2, pp = l15 & 3;
        const int k0 = 8 * g + qd, k1 = k0 + 4;
        const int cidx = (nr >> 3) + (pp >> 1), half = (pp & 1) << 2;
        const int o0 = k0 * BN + ((cidx ^ (k0 & 7)) << 3) + half, o1 = k1 * BN + ((cidx ^ (k1 & 7)) << 3) + half;
        const u16x4 h0 = lds_read_tr16(&bH[o0]), h1 = lds_read_tr16(&bH[o1]);
        const u16x4 l0 = lds_read_tr16(&bL[o0]), l1 = lds_read_tr16(&bL[o1]);
        fbh[j] = u16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        fbl[j] = u16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
      }
    }
  };
  // small terms first, then the leading product (gemm_x3.hip); the two accumulators alternate, so no matrix instruction waits for
  // the one issued right before it
  auto do_mfma = [&](const u16x8& fah, const u16x8& fal, const u16x8 (&fbh)[TN], const u16x8 (&fbl)[TN]) {
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[j] = mfma_bf16_16x16x32(fal, fbh[j], acc[j]);
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[j] = mfma_bf16_16x16x32(fah, fbl[j], acc[j]);
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[j] = mfma_bf16_16x16x32(fah, fbh[j], acc[j]);
  };

  // tile 0 -> set 0; its register slot takes tile 2
  convert(ra0, rb0, 0);
  X3_STAMP(2);
  if (nkt > 2) fetch(ra0, rb0);
  lds_barrier();
  X3_STAMP(3);
  // one k-tile: fragments of set `kt & 1` requested first, then tile kt + 1 converted into the other set (its register slot refilled
  // with tile kt + 3), then the matrix instructions; ONE barrier: every wave is done reading this set and writing the other one
  auto ktile = [&](int kt, int set, f32x4& ra, f32x4& rb) {   // (ra, rb): the slot of tile kt + 1
    if (kt < 40) X3_STAMP(4 + kt);
    u16x8 fah, fal, fbh[TN], fbl[TN];
    load_frags(set, fah, fal, fbh, fbl);
    if (kt + 1 < nkt) convert(ra, rb, set ^ 1);
    if (kt + 3 < nkt) fetch(ra, rb);
    do_mfma(fah, fal, fbh, fbl);
    lds_barrier();
  };
  for (int kt = 0; kt < nkt; kt += 2) {
    ktile(kt, 0, ra1, rb1);
    if (kt + 1 < nkt) ktile(kt + 1, 1, ra0, rb0);
  }
  X3_STAMP(50);
  if (AKS || q.atomic) {
    // ---- split-K epilogue (weight-gradient form; forward form with atomic output): fp32 atomics into C, bias gradient from the
    //      n-tile-0 workgroups ----
    if (nkt <= 0) return;
    const float alpha = q.alpha;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const long n = n0 + wn * 16 * TN + j * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long m = m0 + wm * 16 + 4 * g + r;
        if (m < M && n < N) atomicAdd(q.C + m * q.ldc + n, alpha * acc[j][r]);
      }
    }
    if (AKS && q.colsum && nt == 0) {   // 32 threads (lines) hold partial sums of the same four columns: fold through LDS
      __syncthreads();
      float* red = reinterpret_cast<float*>(smem);   // [32 lines][64 columns]
      st4(red + lnA * 64 + chA * 4, csum);
      __syncthreads();
      if (tid < 64 && m0 + tid < M) {
        float s = 0.f;
#pragma unroll 8
        for (int l = 0; l < 32; ++l) s += red[l * 64 + tid];
        atomicAdd(q.colsum + m0 + tid, s);
      }
    }
    return;
  }
  // residual rows of this thread: requested before the accumulators go through LDS
  const bool has_res = q.res != nullptr;
  float rv[NIT][4];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    rv[it][0] = rv[it][1] = rv[it][2] = rv[it][3] = 0.f;
    const long m = m0 + er0 + it * RSTEP;
    if (has_res && ecol && m < M) {
      const long ri = m * q.ldres + en;
      if (full) ld4(q.res + ri, rv[it]);
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (en + e < N) rv[it][e] = q.res[ri + e];
      }
    }
  }
  __syncthreads();
  float* sC = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) sC[(wm * 16 + 4 * g + r) * PITCH_C + wn * 16 * TN + j * 16 + l15] = acc[j][r];
  __syncthreads();
  X3_STAMP(51);
  if (!ecol) return;
  const float alpha = q.alpha, beta = q.beta;
  const bool has_beta = beta != 0.f, has_rs = q.rowscale != nullptr;
  const int act = q.act;
  // the common epilogue (gemm_lean.hip): whole column quads, plain store, bias (+ residual), nothing else
  if (full && q.c_patch_ow == 0 && !has_rs && !has_beta && act == 0) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = er0 + it * RSTEP;
      const long m = m0 + row;
      if (m >= M) break;
      const float4 t = *reinterpret_cast<const float4*>(&sC[row * PITCH_C + q4]);
      float v[4] = {alpha * t.x + bv[0] + rv[it][0], alpha * t.y + bv[1] + rv[it][1], alpha * t.z + bv[2] + rv[it][2], alpha * t.w + bv[3] + rv[it][3]};
      st4(q.C + m * q.ldc + en, v);
    }
    X3_STAMP(52);
    return;
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int row = er0 + it * RSTEP;
    const long m = m0 + row;
    if (m >= M) break;
    const float4 t = *reinterpret_cast<const float4*>(&sC[row * PITCH_C + q4]);
    float v[4] = {t.x, t.y, t.z, t.w};
    long ci = m * q.ldc + en;
    if (q.c_patch_ow > 0) {   // un-patchify (gemm_kernels.h epilogue_rows): ((boh*KH + kh) * OW + ow) * KW*Ci + (kw*Ci + ci)
      const unsigned boh = (unsigned)m / (unsigned)q.c_patch_ow, ow = (unsigned)m - boh * (unsigned)q.c_patch_ow;
      const unsigned kh = (unsigned)en / (unsigned)q.c_patch_kwci, rest = (unsigned)en - kh * (unsigned)q.c_patch_kwci;
      ci = (((long)boh * q.c_patch_kh + kh) * q.c_patch_ow + ow) * q.c_patch_kwci + rest;
    }
    float ov[4] = {0.f, 0.f, 0.f, 0.f};
    float rs = 1.f;
    if (has_rs) rs = q.rowscale[(unsigned)m / (unsigned)q.rows_per_scale];
    if (has_beta) {
      if (full) ld4(q.C + ci, ov);
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (en + e < N) ov[e] = q.C[ci + e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = alpha * v[e] + bv[e];
      const float a = act == 0 ? x : act == 1 ? epi_act<1>(x) : act == 2 ? epi_act<2>(x) : epi_act<3>(x);
      v[e] = a * rs + rv[it][e] + beta * ov[e];
    }
    if (full) st4(q.C + ci, v);
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (en + e < N) q.C[ci + e] = v[e];
    }
  }
  X3_STAMP(52);
}

template <bool AKS, bool BKS>
__global__ __launch_bounds__(512, 2) void gemm_x3_lean_kernel(X3LeanParams q) {
  x3_lean_body<AKS, BKS>(q, (int)blockIdx.x, (int)blockIdx.z, true);
}

// everything of X3LeanParams that follows from the problem alone (host launcher and the grouped kernel alike); the split fields
// (kt_per, splits, atomic) are the caller's
static __host__ __device__ inline void x3_fill(const GemmParams& p, X3LeanParams& q) {
  constexpr int BM = 64, BN = 64;
  q.A = reinterpret_cast<const float*>(p.A.ptr);
  q.B = reinterpret_cast<const float*>(p.B.ptr);
  q.C = reinterpret_cast<float*>(p.C);
  q.bias = p.bias;
  q.res = reinterpret_cast<const float*>(p.res);
  q.rowscale = p.rowscale;
  q.lda = p.A.ld; q.ldb = p.B.ld; q.ldc = p.ldc; q.ldres = p.ldres;
  q.M = p.M; q.N = p.N; q.nkt = p.K / 32;
  q.tiles_n = (p.N + BN - 1) / BN;
  q.ntile = (int)((long)((p.M + BM - 1) / BM) * q.tiles_n);
  q.rows_per_scale = p.rows_per_scale > 0 ? p.rows_per_scale : 1;
  q.act = p.act;
  q.flags = p.c_vec_ok ? 4 : 0;
  q.alpha = p.alpha; q.beta = p.beta;
  q.colsum = p.colsum;
  q.kt_per = q.nkt;
  q.splits = 1;
  q.p_seg = 0; q.p_img = 0; q.p_ohw = q.p_ow = 1; q.p_row = q.p_col = q.p_jump = 0;
  if (p.A.conv == 2) {
    const GemmView& v = p.A;
    q.p_img = (long)v.H * v.W * v.C;
    q.p_ohw = v.OH * v.OW; q.p_ow = v.OW;
    q.p_row = v.stride * v.W * v.C; q.p_col = v.stride * v.C;
    q.p_seg = v.KW * v.C / 32;
    q.p_jump = (v.W - v.KW) * v.C * 4;
  }
  q.atomic = 0;
  q.c_patch_ow = p.c_patch_ow; q.c_patch_kh = p.c_patch_kh; q.c_patch_kwci = p.c_patch_kwci;
  q.pb_kwc = 0; q.pb_img = 0; q.pb_ohw = q.pb_ow = 1; q.pb_row = q.pb_col = q.pb_line = 0;
  if (p.B.conv == 2) {
    const GemmView& v = p.B;
    q.pb_img = (long)v.H * v.W * v.C;
    q.pb_ohw = v.OH * v.OW; q.pb_ow = v.OW;
    q.pb_row = v.stride * v.W * v.C; q.pb_col = v.stride * v.C;
    q.pb_kwc = v.KW * v.C; q.pb_line = v.W * v.C;
  }
  q.batch2 = p.batch2 > 0 ? p.batch2 : 1;
  q.a_bs = p.A.batch_stride; q.a_b2s = p.A.batch2_stride; q.b_bs = p.B.batch_stride; q.b_b2s = p.B.batch2_stride;
  q.c_bs = p.c_batch_stride; q.c_b2s = p.c_batch2_stride; q.r_bs = p.res_batch_stride; q.r_b2s = p.res_batch2_stride;
}

// GROUPED weight-gradient form: the deferred weight gradients of a backward phase in the split-bf16 mode as ONE grid (cmda_gemm_grouped:
// in round 5's first version each of the ~900 per step was its own 29-us launch, 17 % of that mode's kernel time).  tab: the
// problems (splits resolved by the planner); map: {problem, split * tiles + tile} per workgroup, -1 = padding (gemm_grouped.hip)
__global__ __launch_bounds__(512, 2) void gemm_x3_lean_grouped_kernel(const GemmParams* __restrict__ tab, const int* __restrict__ map) {
  const int prob = map[2 * blockIdx.x], blk = map[2 * blockIdx.x + 1];
  if (prob < 0) return;
  const GemmParams& p = tab[prob];
  X3LeanParams q;
  x3_fill(p, q);
  const int splits = p.splits > 0 ? p.splits : 1;
  q.kt_per = (q.nkt + splits - 1) / splits;
  q.splits = splits;
  q.flags |= splits > 1 ? 32 : 0;
  x3_lean_body<true, true>(q, blk % q.ntile, blk / q.ntile, false);
}

int launch_x3_lean(const GemmParams& p, void* stream) {
  constexpr int BM = 64, BN = 64;
  X3LeanParams q;
  const long tiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  if (tiles > 0x7fffffffL) return CMDA_ERR_SHAPE;
  x3_fill(p, q);
  const long nb = (long)p.batch * q.batch2;
  const dim3 blk(512);
  if (p.a_kstrided) {
    // split-K over the tokens: ~two workgroups per CU in all, at least eight k-tiles per split (one fp32 atomic per output element
    // per split)
    // (tools/dbg/x3_wgrad_bench.py: 256 / 384 / 1024 workgroups and 4 / 16 k-tiles per split measured within +-10 % of this choice)
    int splits = (int)std::max<long>(1, std::min<long>(q.nkt / 8, (512 + tiles * nb - 1) / (tiles * nb)));
    splits = std::min(splits, 1024);
    q.kt_per = (q.nkt + splits - 1) / splits;
    splits = (q.nkt + q.kt_per - 1) / q.kt_per;
    q.splits = splits;
    q.flags |= (nb > 1 ? 16 : 0) | (splits > 1 ? 32 : 0);
    if (nb * splits > 65535) return CMDA_ERR_SHAPE;
    CMDA_LAUNCH((gemm_x3_lean_kernel<true, true>), dim3((unsigned)tiles, 1, (unsigned)(nb * splits)), blk, 0, stream, q);
    CMDA_CHECK_LAUNCH();
  }
  if (p.atomic) {   // forward form, split-K with the caller's / dispatcher's split count
    int splits = std::max(1, std::min(p.splits, q.nkt));
    q.kt_per = (q.nkt + splits - 1) / splits;
    q.splits = (q.nkt + q.kt_per - 1) / q.kt_per;
    q.atomic = 1;
  }
  if (nb * q.splits > 65535) return CMDA_ERR_SHAPE;
  q.flags |= (nb > 1 ? 16 : 0) | (q.splits > 1 ? 32 : 0);
  const dim3 grid((unsigned)tiles, 1, (unsigned)(nb * q.splits));
  if (p.b_kstrided) CMDA_LAUNCH((gemm_x3_lean_kernel<false, true>), grid, blk, 0, stream, q);
  else CMDA_LAUNCH((gemm_x3_lean_kernel<false, false>), grid, blk, 0, stream, q);
  CMDA_CHECK_LAUNCH();
}

}  // namespace

#ifdef CMDA_X3_TIMING
extern "C" int cmda_debug_x3_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_x3_stamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -3;
}
#endif

// HOST: does the lean split-bf16 kernel take this problem?
bool cmda_gemm_x3_lean_ok_(const cmda_gemm_params_t& p) {
  auto plain = [](const GemmView& v) {
    return v.conv == 0 && v.vec_ok && (v.ld % 4) == 0 && v.R < (1L << 31) && v.Cc < (1L << 31) && (reinterpret_cast<uintptr_t>(v.ptr) % 16) == 0;
  };
  const long nb = (long)p.batch * (p.batch2 > 0 ? p.batch2 : 1);
  auto bs_ok = [](const GemmView& v) { return (v.batch_stride % 4) == 0 && (v.batch2_stride % 4) == 0; };
  // patch view of a kernel == stride convolution (no batch): 32-bit row arithmetic, 16-byte pieces
  auto patch_geom = [&](const GemmView& v) {
    return v.conv == 2 && v.vec_ok && v.KH == v.stride && v.KW == v.stride && v.pad == 0 && v.dil == 1 && v.in_dil <= 1 &&
           v.H == v.OH * v.stride && v.W == v.OW * v.stride && v.R < (1L << 31) && (long)v.stride * v.W * v.C * 4 < (1L << 31) &&
           p.K > 0 && nb == 1 && (reinterpret_cast<uintptr_t>(v.ptr) % 16) == 0 && (v.C % 4) == 0;
  };
  // ... as A of the forward form (K-contiguous B): KW * C a multiple of the k-tile
  auto patch_a = [&](const GemmView& v) {
    return patch_geom(v) && !p.a_kstrided && !p.b_kstrided && ((long)v.KW * v.C) % 32 == 0 && p.K == (long)v.KH * v.KW * v.C;
  };
  // ... as the K-strided B of the weight-gradient form (dW of the convolution): a 64-column tile inside one kernel row
  auto patch_b = [&](const GemmView& v) {
    return patch_geom(v) && p.a_kstrided && p.b_kstrided && ((long)v.KW * v.C) % 64 == 0 && p.N == (long)v.KH * v.KW * v.C && p.K == v.R;
  };
  if (!(p.dtype == CMDA_F32X3 && (plain(p.A) || patch_a(p.A)) && (plain(p.B) || patch_b(p.B)) && (p.K % 32) == 0 && p.K >= 32 && p.batch >= 1 && nb <= 65535 &&
        bs_ok(p.A) && bs_ok(p.B) && (p.c_batch_stride % 4) == 0 && (p.c_batch2_stride % 4) == 0 && (nb == 1 || !p.colsum) &&
        p.c_perm_ci == 0 && p.out_f32 && (p.N % 4) == 0 && !(p.tile_hint > 0 && (p.tile_hint & 8192))))   // (bit 13: general kernel, tuning A/B)
    return false;
  if (p.a_kstrided)   // weight-gradient form: dW (+)= dy^T x with fp32 atomics, any split count (the kernel chooses its own)
    return p.b_kstrided && p.atomic && (p.M % 4) == 0 && !p.bias && !p.res && !p.rowscale && p.act == 0 && p.beta == 0.f && p.c_patch_ow == 0 &&
           32L * p.A.ld * 4 < (1L << 31) && (p.B.conv == 2 || 32L * p.B.ld * 4 < (1L << 31));
  if (p.colsum || (p.b_kstrided && 32L * p.B.ld * 4 >= (1L << 31))) return false;
  if (p.atomic)       // forward form with split-K: atomics into pre-filled rows (plain epilogue only; the caller's split count)
    return nb == 1 && p.c_patch_ow == 0 && !p.bias && !p.res && !p.rowscale && p.act == 0 && p.beta == 0.f && p.splits >= 1;
  if (p.c_patch_ow > 0)   // patch-store epilogue (spatial-reduction data gradients): 32-bit row / column arithmetic
    return nb == 1 && !p.res && p.c_vec_ok;
  return p.splits <= 1;
}

int cmda_gemm_x3_lean_(const cmda_gemm_params_t& p, void* stream) { return launch_x3_lean(p, stream); }

// grouped weight-gradient launch (gemm_grouped.hip plans the splits and the block map; tab / blk are DEVICE pointers)
int cmda_gemm_x3_lean_grouped_(const cmda_gemm_params_t* tab, const void* blk, int nblocks, void* stream) {
  if (nblocks <= 0) return CMDA_OK;
  CMDA_LAUNCH(gemm_x3_lean_grouped_kernel, dim3((unsigned)nblocks), dim3(512), 0, stream, tab, reinterpret_cast<const int*>(blk));
  CMDA_CHECK_LAUNCH();
}
