// mixffn.hip -- the MLP half of a MiT block as ONE kernel:  x2 = x1 + DropPath(fc2(GELU(dw3x3(fc1(LayerNorm2(x1))))))
// (mix_transformer.py:20-44 Mlp.forward, :443-455 DWConv, :141-146 Block.forward's second residual branch).
//
// Why: at 2 + 2 samples per GPU the four launches behind this line (LayerNorm, fc1, depthwise + GELU, fc2) are 6 - 17 us each with
// ~1.9 us of dependent-launch boundary in between, every one of them spreads >= 256 small workgroups over the chip and is bound by the
// L2 -> LDS fill rate of a 64 x 64 tile, and the hidden [tokens, 4C] tensor makes three HBM round trips.  Only
// the depthwise stencil couples rows, and only along the image: a workgroup here owns R whole image lines of ONE sample (plus one
// halo line on each side, recomputed), keeps its normalised rows in REGISTERS as MFMA A-fragments for the whole kernel, and walks the
// hidden dimension in sub-chunks of 64 channels:
//     fc1 (128 rows x 64 hidden, K = C)  ->  + b1, bf16, into LDS  ->  3 x 3 stencil + bias + GELU on the R interior lines (LDS -> LDS)
//     ->  fc2 partial (64 rows x C, K = 64) accumulated in registers over all sub-chunks
// so the hidden activations never leave the CU.  The only stream through the LDS-DMA ring is the weights (W1 / W2 in 8 KiB pieces,
// 2 * C / 64 pieces per sub-chunk, three pieces in flight) plus one 3 KiB piece of per-channel parameters (depthwise taps, both
// biases) per sub-chunk.  No hidden split across workgroups: the output rows are complete, the epilogue adds the fp32 residual and
// stores -- no atomics (1.3 TB/s chip-wide would cost more than the kernel), no partial buffers.  The grid is B * ceil(H / R)
// workgroups (64 / 128 for the student's stage-3 passes): deliberately under-filling, so the two encoder lanes of the step overlap
// instead of time-slicing one another.
//
// Rounding points are those of the separate launches (normalised rows, fc1 output and GELU output rounded to bf16, everything else
// fp32); sums associate differently (MFMA k order, 8-lane LayerNorm reduction), so results agree to bf16 / fp32 round-off, not bit
// for bit.
// Training saves (student passes): normalised rows, LayerNorm statistics, fc1 output h and the activation -- what the (unfused)
// backward chain reads.
#include "common.h"
#include <type_traits>

namespace {

__device__ __attribute__((aligned(16))) unsigned g_mf_zero[4] = {0u, 0u, 0u, 0u};

// -DCMDA_MIXFFN_TIMING (tuning builds, tools/dbg/mixffn_phase.py): lane 0 of wave 0 of workgroup 0 accumulates s_memtime deltas per
// phase: 0 prologue (DMA issue + LayerNorm), 1 fragments, 2 fc1 steps (waits included), 3 h epilogue + barrier, 4 stencil + barrier,
// 5 fc2 steps, 6 epilogue, 7 total; 8.. the waits alone (barrier of every fc1 / fc2 step)
#ifdef CMDA_MIXFFN_TIMING
__device__ unsigned long long g_mf_stamps[16];
#define MF_T0() unsigned long long mf_t = __builtin_amdgcn_s_memtime(), mf_acc[16] = {0}; const unsigned long long mf_start = mf_t
#define MF_LAP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); mf_acc[i] += n_ - mf_t; mf_t = n_; } while (0)
#define MF_DUMP() do { if (threadIdx.x == 0 && blockIdx.x == 0) { mf_acc[7] = __builtin_amdgcn_s_memtime() - mf_start; \
    for (int i_ = 0; i_ < 16; ++i_) g_mf_stamps[i_] = mf_acc[i_]; } } while (0)
#ifdef CMDA_MIXFFN_NOWAIT   // timing experiment only (WRONG results): barriers without the DMA waits -> what is left is wave skew
#define pipe_barrier_nowait lds_barrier
#endif
#else
#define MF_T0() do { } while (0)
#define MF_LAP(i) do { } while (0)
#define MF_DUMP() do { } while (0)
#endif

struct MixFfnParams {
  const float* x;          // [M, C] fp32 residual stream
  const float* gamma;      // norm2
  const float* beta;
  const bf16_t* W1;        // [Hd, C]
  const float* b1;         // [Hd]
  const float* wdw;        // [9][Hd] tap-major
  const float* bdw;        // [Hd]
  const bf16_t* W2;        // [C, Hd]
  const float* b2;         // [C]
  const float* rowscale;   // [B] per-sample DropPath factor or null
  float* out;              // [M, C]
  bf16_t* xn;              // saves (each may be null)
  float* mean;
  float* rstd;
  bf16_t* h;
  bf16_t* act;
  int B, H, W, Hd, R, panels;
  float eps;
};

constexpr int MF_PI = 128;            // fc1 rows per workgroup (R + 2 lines, padded)
constexpr int MF_PO = 64;             // output rows per workgroup (R lines, padded)
constexpr int MF_PIECE = 64 * 64;     // elements of a weight piece (64 lines x 64 k)
constexpr int MF_PARAM_F = 12 * 64;   // floats of a parameter piece: rows 0..8 taps, 9 depthwise bias, 10 fc1 bias, 11 unused
constexpr int MF_HROWS = 136;         // rows of the fc1-output image: (R + 2) lines x (W + 2) columns (a zero column on either side)
constexpr int MF_HP = 96;             // its row pitch in elements: 192 bytes -- rows two apart (a wave's four pixel pairs) alternate bank halves
constexpr int MF_ACT_OFF = 26624;     // byte offset of the activation tile behind it

// LDS map: region 0 = the LayerNorm panel of one 64-row pass (prologue), then the fc1-output image + the activation tile; the weight
// ring (two groups of C / 64 pieces: W1 of a sub-chunk, W2 of a sub-chunk); two parameter pieces; the epilogue's fp32 tile overlays all
template <int NKT>
struct MfLds {
  static constexpr size_t REG0 = (size_t)NKT * 64 * 64 * 2 > MF_ACT_OFF + 8192 ? (size_t)NKT * 64 * 64 * 2 : (size_t)35840;
  static constexpr size_t RING = REG0, PARAM = RING + (size_t)2 * NKT * MF_PIECE * 2, END = PARAM + 2 * MF_PARAM_F * 4;
  static constexpr size_t STAGE = (size_t)MF_PO * (NKT * 64 + 4) * 4;
  static constexpr size_t BYTES = END > STAGE ? END : STAGE;
};

template <int NKT>
__global__ __launch_bounds__(512, 1) void mixffn_fwd_kernel(MixFfnParams q) {
  typedef bf16_t T;
  constexpr int C = NKT * 64, NVL = 2 * NKT;
  constexpr int SZ_P = 64 * 64;   // one k-tile of the 64-row LayerNorm panel
  __shared__ __attribute__((aligned(1024))) char smem[MfLds<NKT>::BYTES];
  T* const sPanel = reinterpret_cast<T*>(smem);
  T* const sH = sPanel;                                              // [MF_HROWS][MF_HP] after the prologue
  T* const sAct = reinterpret_cast<T*>(smem + MF_ACT_OFF);           // [64][64], MFMA image
  T* const sW1 = reinterpret_cast<T*>(smem + MfLds<NKT>::RING);      // group 0: NKT pieces [64 hidden lines][64 k]
  T* const sW2 = sW1 + NKT * MF_PIECE;                               // group 1: NKT pieces [64 output lines][64 k]
  float* const sParam = reinterpret_cast<float*>(smem + MfLds<NKT>::PARAM);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const int W = q.W, H = q.H, Hd = q.Hd, R = q.R;
  const int nsub = Hd >> 6;
  const int b = (int)((unsigned)blockIdx.x / (unsigned)q.panels), panel = blockIdx.x - b * q.panels;
  const int y0 = panel * R;
  const int lines_out = min(R, H - y0);
  const int rows_out = lines_out * W;                     // interior rows that exist
  const int rows_in = (R + 2) * W;                        // fc1 rows (incl. halo lines)
  const long obase = ((long)b * H + y0) * W;              // global row of interior row 0
  const long ibase = obase - W;                           // global row of fc1 row 0 (line y0 - 1)

  // ---- weight ring: per-lane source bases (LDS image of the GEMM kernels: line r, 16-byte slot = chunk ^ (r & 7)) ----
  const int ln = wid * 8 + (lane >> 3), chunk = (lane & 7) ^ (ln & 7);
  const T* src1 = q.W1 + (long)ln * C + chunk * 8;                                        // + j * 64 * C + kt * 64
  const T* src2 = q.W2 + ((long)(ln >> 4) * (C / 4) + (ln & 15)) * Hd + chunk * 8;        // + i * 16 * Hd + j * 64
  auto issue_w1 = [&](int j) {   // the NKT k-tiles of W1 rows j * 64 .. + 63
    const T* s = src1 + (long)j * 64 * C;
#pragma unroll
    for (int i = 0; i < NKT; ++i) glds16_asm(s + i * 64, reinterpret_cast<char*>(sW1 + i * MF_PIECE) + wid * 1024);
  };
  auto issue_w2 = [&](int j) {   // W2 columns j * 64 .. + 63: piece i = output channels {wn2 * C / 4 + 16 i .. + 15 : wn2 = 0 .. 3}
    const T* s = src2 + j * 64;
#pragma unroll
    for (int i = 0; i < NKT; ++i) glds16_asm(s + (long)i * 16 * Hd, reinterpret_cast<char*>(sW2 + i * MF_PIECE) + wid * 1024);
  };
  // per-channel parameters of a sub-chunk (nine depthwise taps, depthwise bias, fc1 bias: 11 rows of 64 floats): threads 0 .. 175
  // fetch one 16-byte piece each into a REGISTER early and store it into the other LDS slot a phase later.  (Through the LDS-DMA
  // like the weights the compiler could not tell the stencil's parameter reads from the pending DMA writes and drained the whole
  // vector-memory queue -- s_waitcnt vmcnt(0) -- in front of them: every weight group's latency was exposed.)
  const bool prm_thread = tid < 176;
  auto load_param = [&](int j) -> float4 {   // (every thread loads: rows past 10 re-read the fc1 bias row and are not stored)
    const int row = tid >> 4, c4 = (tid & 15) * 4;
    const float* s = row < 9 ? q.wdw + (long)row * Hd + j * 64 + c4 : row == 9 ? q.bdw + j * 64 + c4 : q.b1 + j * 64 + c4;
    return *reinterpret_cast<const float4*>(s);
  };
  auto store_param = [&](const float4& v, int slot) {
    if (prm_thread) *reinterpret_cast<float4*>(sParam + slot * MF_PARAM_F + tid * 4) = v;
  };
  MF_T0();
  issue_w1(0);
  store_param(load_param(0), 0);

  // ---- LayerNorm of the (R + 2) * W rows, 64 rows per pass, eight lanes per row (gemm_ln.hip); both passes' rows are requested up
  //      front; pass p's panel becomes the A fragments (32 rows x C per wave, registers) of waves 4p .. 4p + 3 ----
  const int wm = wid >> 1, wn = wid & 1;      // fc1 roles: 32-row group, 32 hidden columns
  u16x8 afr[2][NVL];
  {
    const int l8 = tid & 7;
    float v[2][NVL][4];
    bool live[2], interior[2];
    long grow[2];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int prow = pass * 64 + (tid >> 3);
      const int ll = (int)((unsigned)prow / (unsigned)W);
      const int y = y0 - 1 + ll;
      live[pass] = prow < rows_in && y >= 0 && y < H;
      interior[pass] = live[pass] && ll >= 1 && ll <= lines_out;
      grow[pass] = live[pass] ? ibase + prow : obase;
      const float* xr = q.x + grow[pass] * C;
#pragma unroll
      for (int i = 0; i < NVL; ++i) ld4(xr + (i * 8 + l8) * 4, v[pass][i]);
    }
    float gm[NVL][4], bt4[NVL][4];   // requested with the rows (inside the loop below each pair cost its own L2 round trip)
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      ld4(q.gamma + (i * 8 + l8) * 4, gm[i]);
      ld4(q.beta + (i * 8 + l8) * 4, bt4[i]);
    }
    MF_LAP(11);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int prow = tid >> 3;     // row of this pass's panel
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NVL; ++i) s += (v[pass][i][0] + v[pass][i][1]) + (v[pass][i][2] + v[pass][i][3]);
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      const float mean = s / (float)C;
      float qs = 0.f;
#pragma unroll
      for (int i = 0; i < NVL; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[pass][i][e] - mean;
          qs += d * d;
        }
      qs += __shfl_xor(qs, 1, 64);
      qs += __shfl_xor(qs, 2, 64);
      qs += __shfl_xor(qs, 4, 64);
      const float rstd = rsqrtf(qs / (float)C + q.eps);
      MF_LAP(12);
#pragma unroll
      for (int i = 0; i < NVL; ++i) {
        const int vi = i * 8 + l8;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = live[pass] ? (v[pass][i][e] - mean) * rstd * gm[i][e] + bt4[i][e] : 0.f;
        T* dst = sPanel + (vi >> 4) * SZ_P + prow * 64 + (((((vi & 15) >> 1)) ^ (prow & 7)) << 3) + ((vi & 1) << 2);
        st4(dst, o);
        if (interior[pass] && q.xn) st4(q.xn + grow[pass] * C + vi * 4, o);
      }
      if (interior[pass] && l8 == 0 && q.mean) {
        q.mean[grow[pass]] = mean;
        q.rstd[grow[pass]] = rstd;
      }
      MF_LAP(13);
      __syncthreads();
      MF_LAP(14);
      if ((wid >> 2) == pass) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int row = (wm & 1) * 32 + mi * 16 + l15;
#pragma unroll
          for (int ks = 0; ks < NVL; ++ks)
            afr[mi][ks] = *reinterpret_cast<const u16x8*>(&sPanel[(ks >> 1) * SZ_P + row * 64 + ((((ks & 1) * 4 + g) ^ (row & 7)) << 3)]);
        }
      }
      __syncthreads();
    }
  }
  MF_LAP(0);
  // the fc1-output image lives where the panel was: zero its two padding columns once (the stencil's left / right border)
  for (int i = tid; i < (R + 2) * 2 * 12; i += 512) {
    const int rr = i / 12, c16 = i - rr * 12;
    const int row = (rr >> 1) * (W + 2) + ((rr & 1) ? W + 1 : 0);
    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(sH + row * MF_HP) + c16 * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
  // this lane's two fc1 rows: their row of the image (-1: a padding row of the 128-row tile) and whether they lie inside the sample
  // (rows of lines outside it hold h = 0: the convolution's zero padding)
  int hrow[2];
  bool hrow_ok[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int row = wm * 32 + mi * 16 + l15;
    const int ll = (int)((unsigned)row / (unsigned)W), y = y0 - 1 + ll;
    hrow[mi] = row < rows_in ? ll * (W + 2) + (row - ll * W) + 1 : -1;
    hrow_ok[mi] = y >= 0 && y < H;
  }
  // stencil role: channel quad cq, pixel pair (p0, p0 + 1) of the interior
  const int cq = tid & 15, p0 = (tid >> 4) * 2;
  const bool st_live = p0 < rows_out;
  const int st_l = (int)((unsigned)p0 / (unsigned)W), st_x = p0 - st_l * W;
  const T* const st_base = sH + (st_l * (W + 2) + st_x) * MF_HP + cq * 4;   // window column x - 1 of window line 0
  const int st_line = (W + 2) * MF_HP;
  // fc2 roles: wm2 (32 output rows), wn2 (C / 4 output channels)
  const int wm2 = wid & 1, wn2 = wid >> 1;
  f32x4 acc2[2][NKT];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int t = 0; t < NKT; ++t) acc2[mi][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  MF_LAP(1);

  auto sub_chunk = [&](const int j, auto slot_tag) {
    constexpr int PS = decltype(slot_tag)::value;
    const float* prm = sParam + PS * MF_PARAM_F;
    // W1 of this sub-chunk has landed (and, j > 0, everyone is done with W2 of the last one): request W2 of this one
#ifdef CMDA_MIXFFN_NOWAIT
    lds_barrier();
#else
    pipe_barrier<0>();
#endif
    issue_w2(j);
    MF_LAP(8);
    // ---- fc1, operands swapped: D[hidden channel][row] -- a lane ends up with four ADJACENT channels of one row ----
    f32x4 acc1[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc1[mi][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const T* sB = sW1 + kt * MF_PIECE;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        u16x8 fb[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int rb = wn * 32 + t * 16 + l15;
          fb[t] = *reinterpret_cast<const u16x8*>(&sB[rb * 64 + (((kk * 4 + g) ^ (rb & 7)) << 3)]);
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int t = 0; t < 2; ++t) acc1[mi][t] = mfma_bf16_16x16x32(fb[t], afr[mi][2 * kt + kk], acc1[mi][t]);
      }
    }
    MF_LAP(2);
    // ---- h = fc1 + b1 (zero on lines outside the sample), bf16, into the image ----
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int c0 = wn * 32 + t * 16 + 4 * g;
      float bb[4];
      ld4(prm + 10 * 64 + c0, bb);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = hrow_ok[mi] ? acc1[mi][t][e] + bb[e] : 0.f;
        if (hrow[mi] >= 0) st4(sH + hrow[mi] * MF_HP + c0, o);
      }
    }
    MF_LAP(10);
    lds_barrier();    // the image is complete; every wave is done reading W1
    float4 next_prm = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j + 1 < nsub) {
      next_prm = load_param(j + 1);   // (requested ahead of the weights: waiting for it leaves their pieces in flight)
      issue_w1(j + 1);
    }
    MF_LAP(3);
    // ---- depthwise 3 x 3 + bias + GELU on the interior lines: thread = (channel quad, two adjacent pixels) ----
    {
      float o0[4] = {0.f, 0.f, 0.f, 0.f}, o1[4] = {0.f, 0.f, 0.f, 0.f};
      if (st_live) {
        float bs[4];
        ld4(prm + 9 * 64 + cq * 4, bs);
#pragma unroll
        for (int e = 0; e < 4; ++e) o0[e] = o1[e] = bs[e];
        // one window line at a time (4 columns x 4 channels live): line kh of both pixels' windows, taps kh * 3 + kw
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          float win[4][4];                  // [column x - 1 .. x + 2][channel]
#pragma unroll
          for (int cx = 0; cx < 4; ++cx) ld4(st_base + kh * st_line + cx * MF_HP, win[cx]);
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            float wv[4];
            ld4(prm + (kh * 3 + kw) * 64 + cq * 4, wv);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              o0[e] += win[kw][e] * wv[e];
              o1[e] += win[kw + 1][e] * wv[e];
            }
          }
          if (kh == 1 && q.h) {   // the fc1 output of the two pixels (the unfused backward recomputes the pre-activation from it)
            st4(q.h + (obase + p0) * Hd + j * 64 + cq * 4, win[1]);
            st4(q.h + (obase + p0 + 1) * Hd + j * 64 + cq * 4, win[2]);
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o0[e] = gelu_erf(o0[e]);
          o1[e] = gelu_erf(o1[e]);
        }
        if (q.act) {
          st4(q.act + (obase + p0) * Hd + j * 64 + cq * 4, o0);
          st4(q.act + (obase + p0 + 1) * Hd + j * 64 + cq * 4, o1);
        }
      }
      st4(sAct + p0 * 64 + ((((cq >> 1)) ^ (p0 & 7)) << 3) + ((cq & 1) << 2), o0);
      st4(sAct + (p0 + 1) * 64 + ((((cq >> 1)) ^ ((p0 + 1) & 7)) << 3) + ((cq & 1) << 2), o1);
    }
    MF_LAP(4);
    // the activation tile is complete and W2 has landed (younger in this wave's queue: the NKT pieces of the next W1, at least)
#ifdef CMDA_MIXFFN_NOWAIT
    lds_barrier();
#else
    if (j + 1 < nsub) pipe_barrier<NKT>();
    else pipe_barrier<0>();
#endif
    MF_LAP(9);
    // ---- fc2 partial: 64 rows x C over this sub-chunk's 64 channels ----
    {
      u16x8 a2[2][2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int row = wm2 * 32 + mi * 16 + l15;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          a2[mi][kk] = *reinterpret_cast<const u16x8*>(&sAct[row * 64 + (((kk * 4 + g) ^ (row & 7)) << 3)]);
      }
      const int rb = wn2 * 16 + l15;
#pragma unroll
      for (int i = 0; i < NKT; ++i) {
        const T* sB = sW2 + i * MF_PIECE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const u16x8 fb = *reinterpret_cast<const u16x8*>(&sB[rb * 64 + (((kk * 4 + g) ^ (rb & 7)) << 3)]);
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) acc2[mi][i] = mfma_bf16_16x16x32(a2[mi][kk], fb, acc2[mi][i]);
        }
      }
    }
    // the next sub-chunk's parameters go into the other slot HERE: the compiler waits for their load with vmcnt(0) (it does not see
    // the DMA pieces in the queue), and the barrier that follows needs the queue empty anyway
    // (the empty statement makes every lane consume the load here, on every path: left to the conditional store alone, the
    // compiler parks a second vmcnt(0) behind the next DMA issue for the lanes that skipped it -- a register-reuse hazard)
#ifndef CMDA_EMU
    asm volatile("" : "+v"(next_prm.x), "+v"(next_prm.y), "+v"(next_prm.z), "+v"(next_prm.w));
#endif
    if (j + 1 < nsub) store_param(next_prm, 1 - PS);
    MF_LAP(5);
  };
  for (int j = 0; j < nsub; j += 2) {
    sub_chunk(j, std::integral_constant<int, 0>{});
    if (j + 1 < nsub) sub_chunk(j + 1, std::integral_constant<int, 1>{});
  }

  // ---- epilogue: accumulators -> fp32 LDS tile -> out = x + rowscale * (y + b2), 16-byte rows ----
  __syncthreads();
  constexpr int PITCH = C + 4;
  float* sC = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        sC[(wm2 * 32 + mi * 16 + 4 * g + r) * PITCH + wn2 * (C / 4) + t * 16 + l15] = acc2[mi][t][r];
  __syncthreads();
  const float rs = q.rowscale ? q.rowscale[b] : 1.f;
  constexpr int QPR = C / 4;
  for (int idx = tid; idx < MF_PO * QPR; idx += 512) {
    const int row = idx / QPR, c4 = (idx - row * QPR) * 4;
    if (row >= rows_out) break;
    const float4 t = *reinterpret_cast<const float4*>(&sC[row * PITCH + c4]);
    float xv[4], bv[4], o[4];
    const long gi = (obase + row) * C + c4;
    ld4(q.x + gi, xv);
    ld4(q.b2 + c4, bv);
    o[0] = xv[0] + rs * (t.x + bv[0]);
    o[1] = xv[1] + rs * (t.y + bv[1]);
    o[2] = xv[2] + rs * (t.z + bv[2]);
    o[3] = xv[3] + rs * (t.w + bv[3]);
    st4(q.out + gi, o);
  }
  MF_LAP(6);
  MF_DUMP();
}

static inline int mixffn_lines(int W, int R_hint) {
  int R = std::min(MF_PI / W - 2, MF_PO / W);
  if (R_hint > 0) R = std::min(R, R_hint);
  return R;
}

}  // namespace

#ifdef CMDA_MIXFFN_TIMING
extern "C" int cmda_debug_mixffn_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_mf_stamps), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -3;
}
#endif

// HOST: does the fused kernel take this MixFFN?  (bf16 compute with the fp32 residual stream; C a multiple of 64 up to 320; even
// image width up to 42 so that at least one line + its two halo lines fit the 128-row panel)
extern "C" int cmda_mixffn_fwd_ok(int B, int H, int W, int C, int Hd) {
  if (B <= 0 || H <= 0 || W < 2 || (W & 1) || C % 64 != 0 || C < 128 || C > 320 || Hd % 64 != 0 || Hd < 64) return 0;
  if (mixffn_lines(W, 0) < 1) return 0;
  if ((long)B * H * W * (long)std::max(C, Hd) >= (1L << 31)) return 0;
  return 1;
}

// x2 = x + rowscale[b] * (fc2(GELU(dw3x3(fc1(LN(x)) ) + bdw)) + b2).  x / out fp32 [B*H*W, C]; w1 [Hd, C], w2 [C, Hd] bf16;
// wdw tap-major fp32 [9][Hd].  Saves (bf16 xn [M, C], fp32 mean / rstd [M], bf16 h / act [M, Hd]) are written when non-null.
// lines_hint > 0 caps the image lines per workgroup (more, smaller workgroups for short passes).
extern "C" int cmda_mixffn_fwd(const float* x, const float* gamma, const float* beta, float eps, const void* w1, const float* b1,
                               const float* wdw, const float* bdw, const void* w2, const float* b2, const float* rowscale, float* out,
                               void* xn, float* mean, float* rstd, void* h, void* act, int B, int H, int W, int C, int Hd,
                               int lines_hint, void* stream) {
  if (!x || !gamma || !beta || !w1 || !b1 || !wdw || !bdw || !w2 || !b2 || !out) return CMDA_ERR_SHAPE;
  if (!cmda_mixffn_fwd_ok(B, H, W, C, Hd)) return CMDA_ERR_UNSUPPORTED;
  if ((mean == nullptr) != (rstd == nullptr)) return CMDA_ERR_SHAPE;
  MixFfnParams q;
  q.x = x; q.gamma = gamma; q.beta = beta;
  q.W1 = reinterpret_cast<const bf16_t*>(w1); q.b1 = b1; q.wdw = wdw; q.bdw = bdw;
  q.W2 = reinterpret_cast<const bf16_t*>(w2); q.b2 = b2; q.rowscale = rowscale; q.out = out;
  q.xn = reinterpret_cast<bf16_t*>(xn); q.mean = mean; q.rstd = rstd;
  q.h = reinterpret_cast<bf16_t*>(h); q.act = reinterpret_cast<bf16_t*>(act);
  q.B = B; q.H = H; q.W = W; q.Hd = Hd;
  q.R = mixffn_lines(W, lines_hint);
  q.panels = (H + q.R - 1) / q.R;
  q.eps = eps;
  const dim3 grid((unsigned)(B * q.panels)), blk(512);
  switch (C / 64) {
    case 2: CMDA_LAUNCH((mixffn_fwd_kernel<2>), grid, blk, 0, stream, q); break;
    case 3: CMDA_LAUNCH((mixffn_fwd_kernel<3>), grid, blk, 0, stream, q); break;
    case 4: CMDA_LAUNCH((mixffn_fwd_kernel<4>), grid, blk, 0, stream, q); break;
    default: CMDA_LAUNCH((mixffn_fwd_kernel<5>), grid, blk, 0, stream, q); break;
  }
  CMDA_CHECK_LAUNCH();
}
