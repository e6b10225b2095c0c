// layernorm.hip -- row LayerNorm over the channel dim of NLC activations.
//
// Reference ops replaced: nn.LayerNorm in mmseg/models/backbones/mix_transformer.py:123,136
// (Block norm1/norm2, eps 1e-6), :175 (OverlapPatchEmbed.norm, eps 1e-5), :76 (Attention.norm,
// eps 1e-5), :270-318 (per-stage norm, eps 1e-6).
//
// HBM-bound: a row is owned by 16, 32 or 64 lanes of a wave (so C = 64 / 128 rows of the first MiT stages still fill
// every lane: 4 / 2 rows per wave), 16-byte (f32) / 8-byte (bf16) lane accesses, xor-shuffle reductions inside the lane
// group, statistics in fp32.  Algorithmic bytes: fwd 2*rows*C*sizeof(T); bwd 3*rows*C*sizeof(T)
// (+ rows*C*sizeof(T) when a residual gradient is fused in).
#include "common.h"
#include <cstdlib>

namespace {

constexpr int kMaxVec = 4;  // 4 * 64 lanes * 4 elems = C up to 1024
constexpr int kSlots = 64;  // partial-sum rows of the backward workspace

// lanes per row: the smallest of 16/32/64 that covers C/4 vectors in one pass (64 otherwise)
static inline int lanes_per_row(int C) {
  const int nvec = C >> 2;
  return nvec <= 16 ? 16 : nvec <= 32 ? 32 : 64;
}
static __device__ __forceinline__ float group_sum(float v, int lpr) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    if (o < lpr) v += __shfl_xor(v, o, 64);
  return v;
}

// TX / TY: storage types of the input and the output.  They differ on the fp32 RESIDUAL STREAM of the bf16 mode (runtime.py
// residual_fp32: the block input x stays fp32 across the 52 residual additions of an encoder, every LayerNorm reads it and hands
// bf16 to the GEMM behind it; the patch-embed norm turns the bf16 convolution output into the fp32 stream).
template <typename TX, typename TY, int NV>
__global__ void ln_fwd_kernel(const TX* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                              TY* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out, long rows,
                              int C, float eps, int lpr) {
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int wpb = blockDim.x >> 6;
  const int rpw = 64 / lpr, li = lane & (lpr - 1), grp = lane / lpr;
  const int nvec = C >> 2;
  // the loop bound is wave-uniform (all lanes take part in the shuffles); a group past the end is only predicated off
  for (long row0 = ((long)blockIdx.x * wpb + wid) * rpw; row0 < rows; row0 += (long)gridDim.x * wpb * rpw) {
    const long row = row0 + grp;
    const bool live = row < rows;
    const TX* xr = x + row * C;
    float v[NV][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = i * lpr + li;
      if (live && vi < nvec) {
        ld4(xr + vi * 4, v[i]);
      } else {
        v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
      }
    }
    // gamma / beta requested WITH the row: behind the two reductions (where they are used) each row paid a second exposed L2 round
    // trip in a kernel that holds one to four rows per wave (round 5; the same finding as the fused-MixFFN prologue)
    float gv[NV][4], bvv[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = i * lpr + li;
      if (vi < nvec) {
        ld4(gamma + vi * 4, gv[i]);
        ld4(beta + vi * 4, bvv[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = group_sum(s, lpr) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = i * lpr + li;
      if (vi < nvec) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = v[i][j] - mean;
          q += d * d;
        }
      }
    }
    const float var = group_sum(q, lpr) / (float)C;
    const float rstd = rsqrtf(var + eps);
    if (!live) continue;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = i * lpr + li;
      if (vi < nvec) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * gv[i][j] + bvv[i][j];
        st4(y + row * C + vi * 4, o);
      }
    }
    if (li == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
  }
}

// dx = [dres +] rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
// per-block partial sums of dy*xhat / dy are added (fp32 atomics) into ws[block % 64][2][C]; ln_bwd_finalize folds the 64
// slots into dgamma / dbeta.  (All blocks hammering the same 2*C addresses was the bottleneck of the first version, one
// workspace row per block made the finalize pass as expensive as the main kernel -- 64 slots keeps both cheap: <= 32
// adds per address, and the grid can be as wide as the rows want.)
// NV = vector passes per lane (ceil(C/4 / lanes per row)): a compile-time bound, so that C = 320 (2 passes) does not carry
// the registers of the C = 1024 case (4 passes) -- 128 VGPRs / 4 waves per SIMD before, and the kernel is latency-bound
// out_scale / dx_scaled (optional): a second output dx * out_scale[row / rows_per_scale] -- the per-sample DropPath factor of the
// residual branch that consumes this gradient next (timm DropPath, mix_transformer.py:145-146), so that no separate scaling
// kernel runs.
template <typename T, typename TX, int NV>
__global__ void ln_bwd_kernel(const T* __restrict__ dy, const TX* __restrict__ x, const float* __restrict__ gamma,
                              const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                              const T* __restrict__ dres, T* __restrict__ dx, float* __restrict__ ws, long rows,
                              int C, int lpr, const float* __restrict__ out_scale, long rows_per_scale,
                              T* __restrict__ dx_scaled) {
  __shared__ float red[2][1024 * NV];  // [gamma/beta][slot = wave*rows_per_wave + group][channel]  (slots*C <= 1024*NV)
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int wpb = blockDim.x >> 6;
  const int rpw = 64 / lpr, li = lane & (lpr - 1), grp = lane / lpr;
  const int nvec = C >> 2;
  float ag[NV][4], ab[NV][4], gm[NV][4];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int vi = i * lpr + li;
#pragma unroll
    for (int j = 0; j < 4; ++j) ag[i][j] = ab[i][j] = gm[i][j] = 0.f;
    if (vi < nvec) ld4(gamma + vi * 4, gm[i]);
  }

  for (long row0 = ((long)blockIdx.x * wpb + wid) * rpw; row0 < rows; row0 += (long)gridDim.x * wpb * rpw) {
    const long row = row0 + grp;
    const bool live = row < rows;
    float mean = 0.f, rstd = 0.f;
    if (live) {
      mean = mean_in[row];
      rstd = rstd_in[row];
    }
    float xh[NV][4], g[NV][4], rr[NV][4];
    float s1 = 0.f, s2 = 0.f;
    // every load of the row is issued before the first use (the residual gradient and the DropPath factor used to be requested
    // behind the row reduction: a second exposed load latency per row in a kernel that holds one row per wave)
    float sc = 1.f;
    if (dx_scaled && live) sc = out_scale[(unsigned)row / (unsigned)rows_per_scale];   // (rows < 2^31: checked by the launcher)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = i * lpr + li;
      rr[i][0] = rr[i][1] = rr[i][2] = rr[i][3] = 0.f;
      if (live && vi < nvec && dres) ld4(dres + row * C + vi * 4, rr[i]);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = i * lpr + li;
      if (live && vi < nvec) {
        float xv[4], dv[4];
        ld4(x + row * C + vi * 4, xv);
        ld4(dy + row * C + vi * 4, dv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xh[i][j] = (xv[j] - mean) * rstd;
          g[i][j] = dv[j] * gm[i][j];
          s1 += g[i][j];
          s2 += g[i][j] * xh[i][j];
          ag[i][j] += dv[j] * xh[i][j];
          ab[i][j] += dv[j];
        }
      }
    }
    s1 = group_sum(s1, lpr) / (float)C;
    s2 = group_sum(s2, lpr) / (float)C;
    if (!live) continue;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = i * lpr + li;
      if (vi < nvec) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = rstd * (g[i][j] - s1 - xh[i][j] * s2);
        if (dres) {
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] += rr[i][j];
        }
        st4(dx + row * C + vi * 4, o);
        if (dx_scaled) {
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] *= sc;
          st4(dx_scaled + row * C + vi * 4, o);
        }
      }
    }
  }
  // cross-group / cross-wave reduce of the parameter gradients, one partial per channel per block
  const int slot = wid * rpw + grp, nslots = wpb * rpw;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int vi = i * lpr + li;
    if (vi < nvec) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        red[0][slot * C + vi * 4 + j] = ag[i][j];
        red[1][slot * C + vi * 4 + j] = ab[i][j];
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float sg = 0.f, sb = 0.f;
    for (int w = 0; w < nslots; ++w) {
      sg += red[0][w * C + c];
      sb += red[1][w * C + c];
    }
    atomicAdd(&ws[((long)(blockIdx.x % kSlots) * 2 + 0) * C + c], sg);
    atomicAdd(&ws[((long)(blockIdx.x % kSlots) * 2 + 1) * C + c], sb);
  }
}

// block = 8 channels x 32 partial-sum lanes: every lane adds nblocks/32 partials, LDS tree over the 32 lanes
__global__ void ln_bwd_finalize_kernel(float* __restrict__ ws, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       int nblocks, int C) {
  __shared__ float red[2][32][8];
  const int cx = threadIdx.x & 7, part = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cx;
  float sg = 0.f, sb = 0.f;
  if (c < C)
    for (int b = part; b < nblocks; b += 32) {
      sg += ws[((long)b * 2 + 0) * C + c];
      sb += ws[((long)b * 2 + 1) * C + c];
      ws[((long)b * 2 + 0) * C + c] = 0.f;  // leave the workspace zeroed for the next call
      ws[((long)b * 2 + 1) * C + c] = 0.f;
    }
  red[0][part][cx] = sg;
  red[1][part][cx] = sb;
  __syncthreads();
  if (part == 0 && c < C) {
    float a = 0.f, b2 = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) { a += red[0][k][cx]; b2 += red[1][k][cx]; }
    dgamma[c] += a;
    dbeta[c] += b2;
  }
}

// Deferred form: MANY LayerNorm layers' workspaces folded by ONE launch at the end of a backward pass (the parameter gradients
// are only read by the optimizer; ~700 finalize launches per UDA step otherwise).  desc[i] = one layer; grid.x = layer,
// grid.y = channel chunks of 256.
struct LnFoldDesc { float* ws; float* dgamma; float* dbeta; int C; int nslots; };
__global__ void ln_fold_batch_kernel(const LnFoldDesc* __restrict__ desc) {
  const LnFoldDesc D = desc[blockIdx.x];
  const int c = blockIdx.y * blockDim.x + threadIdx.x;
  if (c >= D.C) return;
  float sg = 0.f, sb = 0.f;
  for (int k = 0; k < D.nslots; ++k) {
    float* a = D.ws + ((long)k * 2 + 0) * D.C + c;
    float* b = D.ws + ((long)k * 2 + 1) * D.C + c;
    sg += *a;
    sb += *b;
    *a = 0.f;  // leave the workspace zeroed for the next pass
    *b = 0.f;
  }
  D.dgamma[c] += sg;
  D.dbeta[c] += sb;
}

}  // namespace

extern "C" int cmda_layernorm_fwd2(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype, float* mean,
                                   float* rstd, int64_t rows, int C, float eps, void* stream) {
  if (rows <= 0) return CMDA_OK;
  if (C <= 0 || (C & 3) || C > kMaxVec * 256) return CMDA_ERR_SHAPE;
  // C = 320 (80 vectors): 16 lanes x 5 vectors per row, four rows per wave -- every lane busy, four shuffle steps per reduction
  // (64 lanes x 2 passes left 48 lanes idle in the second pass and took six steps).  Forward only: the backward's parameter-gradient
  // fold measured slower on 16-lane groups (ln_bwd_grid).
  const int wpb = 4, lpr = (C == 320 && !std::getenv("CMDA_LN_FWD_WIDE")) ? 16 : lanes_per_row(C);
  const long rpb = wpb * (64 / lpr);
  const int grid = (int)std::min<long>((rows + rpb - 1) / rpb, 8192);
  const int nv = ((C >> 2) + lpr - 1) / lpr;
#define CMDA_LN_FWD_T(TX, TY, NVV)                                                                                          \
  CMDA_LAUNCH((ln_fwd_kernel<TX, TY, NVV>), dim3(grid), dim3(64 * wpb), 0, stream, (const TX*)x, gamma, beta, (TY*)y, mean, \
              rstd, (long)rows, C, eps, lpr)
#define CMDA_LN_FWD(NVV)                                                                         \
  do {                                                                                           \
    if (x_dtype == CMDA_F32 && y_dtype == CMDA_F32) CMDA_LN_FWD_T(float, float, NVV);            \
    else if (x_dtype == CMDA_BF16 && y_dtype == CMDA_BF16) CMDA_LN_FWD_T(bf16_t, bf16_t, NVV);   \
    else if (x_dtype == CMDA_F32 && y_dtype == CMDA_BF16) CMDA_LN_FWD_T(float, bf16_t, NVV);     \
    else if (x_dtype == CMDA_BF16 && y_dtype == CMDA_F32) CMDA_LN_FWD_T(bf16_t, float, NVV);     \
    else return CMDA_ERR_DTYPE;                                                                  \
  } while (0)
  if (nv <= 1) { CMDA_LN_FWD(1); }
  else if (nv == 2) { CMDA_LN_FWD(2); }
  else if (nv == 5) { CMDA_LN_FWD(5); }
  else { CMDA_LN_FWD(4); }
#undef CMDA_LN_FWD
#undef CMDA_LN_FWD_T
  CMDA_CHECK_LAUNCH();
}

extern "C" int cmda_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                  float* rstd, int64_t rows, int C, float eps, int dtype, void* stream) {
  return cmda_layernorm_fwd2(x, dtype, gamma, beta, y, dtype, mean, rstd, rows, C, eps, stream);
}

static inline long ln_bwd_grid(long rows, int C) {
  const long rpb = 4 * (64 / lanes_per_row(C));
  // measured per shape (tools/ln_bench.py under rocprofv3): caps 256 / 512 / 2048 are all slower than 1024; so were 16-lane
  // row groups for C = 320 with a shuffle fold of the parameter gradients (32-38 us against 21.5) and, round 5, with the LDS fold of
  // this kernel (graph-timed 4096 x 320: 23.2 us against 9.8 -- sixteen slots of 320 channels to fold per block)
  const long cap = 1024;
  return std::max<long>(1, std::min<long>((rows + rpb - 1) / rpb, cap));
}

// dgamma / dbeta are ACCUMULATED into (caller zeroes them once per optimizer step).  ws: scratch of
// cmda_layernorm_bwd_ws_floats(rows, C) floats that must be ZERO on entry and is zero again when the call completes (the
// finalize pass clears what it consumed), so one persistent buffer serves every call on a stream without a memset.
extern "C" int64_t cmda_layernorm_bwd_ws_floats(int64_t rows, int C) { (void)rows; return (int64_t)kSlots * 2 * C; }

extern "C" int cmda_layernorm_bwd2(const void* dy, const void* x, int x_dtype, const float* gamma, const float* mean,
                                   const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta, float* ws,
                                   int64_t rows, int C, const float* out_scale, int64_t rows_per_scale, void* dx_scaled,
                                   int dtype, void* stream) {
  if (rows <= 0) return CMDA_OK;
  if (C <= 0 || (C & 3) || C > kMaxVec * 256 || (dx_scaled && (!out_scale || rows_per_scale <= 0))) return CMDA_ERR_SHAPE;
  if (dx_scaled && (rows >= (1L << 31) || rows_per_scale >= (1L << 31))) return CMDA_ERR_SHAPE;   // (32-bit row / rows_per_scale in the kernel)
  const int wpb = 4, lpr = lanes_per_row(C);
  int grid = (int)ln_bwd_grid(rows, C);
  const int nv = ((C >> 2) + lpr - 1) / lpr;
#define CMDA_LN_BWD_T(T, TX, NVV)                                                                                             \
  CMDA_LAUNCH((ln_bwd_kernel<T, TX, NVV>), dim3(grid), dim3(64 * wpb), 0, stream, (const T*)dy, (const TX*)x, gamma, mean, rstd, \
              (const T*)dres, (T*)dx, ws, (long)rows, C, lpr, out_scale, (long)rows_per_scale, (T*)dx_scaled)
#define CMDA_LN_BWD(NVV)                                                                      \
  do {                                                                                        \
    if (dtype == CMDA_F32 && x_dtype == CMDA_F32) CMDA_LN_BWD_T(float, float, NVV);           \
    else if (dtype == CMDA_BF16 && x_dtype == CMDA_BF16) CMDA_LN_BWD_T(bf16_t, bf16_t, NVV);  \
    else if (dtype == CMDA_BF16 && x_dtype == CMDA_F32) CMDA_LN_BWD_T(bf16_t, float, NVV);    \
    else return CMDA_ERR_DTYPE;                                                               \
  } while (0)
  if (nv <= 1) { CMDA_LN_BWD(1); }
  else if (nv == 2) { CMDA_LN_BWD(2); }
  else { CMDA_LN_BWD(4); }
#undef CMDA_LN_BWD
#undef CMDA_LN_BWD_T
  // dgamma == NULL: deferred -- the partial sums stay in `ws` (a workspace owned by this layer) until cmda_layernorm_fold_batch
  if (dgamma)
    CMDA_LAUNCH(ln_bwd_finalize_kernel, dim3((C + 7) / 8), dim3(256), 0, stream, ws, dgamma, dbeta, std::min(grid, kSlots), C);
  CMDA_CHECK_LAUNCH();
}

extern "C" int cmda_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                                  const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta, float* ws,
                                  int64_t rows, int C, const float* out_scale, int64_t rows_per_scale, void* dx_scaled,
                                  int dtype, void* stream) {
  return cmda_layernorm_bwd2(dy, x, dtype, gamma, mean, rstd, dres, dx, dgamma, dbeta, ws, rows, C, out_scale, rows_per_scale,
                             dx_scaled, dtype, stream);
}

// desc: DEVICE array of n records {float* ws; float* dgamma; float* dbeta; int32 C; int32 nslots} (24 bytes + padding to 32):
// dgamma / dbeta += the sums over the nslots partial rows of ws ([nslots][2][C]), ws zeroed.  max_c = the largest C.
extern "C" int cmda_layernorm_fold_batch(const void* desc, int n, int max_c, void* stream) {
  if (n <= 0) return CMDA_OK;
  static_assert(sizeof(LnFoldDesc) == 32, "descriptor layout is part of the ABI");
  CMDA_LAUNCH(ln_fold_batch_kernel, dim3(n, (max_c + 255) / 256), dim3(256), 0, stream, (const LnFoldDesc*)desc);
  CMDA_CHECK_LAUNCH();
}

extern "C" int cmda_layernorm_slots(void) { return kSlots; }
