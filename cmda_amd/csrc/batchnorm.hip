// batchnorm.hip -- train-mode BatchNorm2d (+ReLU) on NHWC activations viewed as [M = B*H*W, C].
//
// Reference: mmcv ConvModule (conv -> BN -> ReLU) instances of the DAFormer head:
// decode_heads/daformer_head.py:46-62 (ASPPWrapper.bottleneck), aspp_head.py:33-43, sep_aspp_head.py:18-27;
// norm_cfg = dict(type='BN') configs/_base_/models/daformer_conv1_mitb5.py:5.  The teacher keeps BN in train mode
// (dacs.py:458-462), so batch statistics + running-stat updates are always on the hot path.
// torch semantics: normalise with the biased batch variance, eps inside the sqrt; running_var uses the unbiased
// variance; momentum 0.1.
//
// Column reductions over M rows: block = 64 channel-quads x 4 row lanes, LDS reduce, one fp32 atomic per channel per
// block.  Variance is accumulated around a per-channel shift (row 0) so the single pass is not cancellation-prone.
// HBM-bound: stats 1 read, apply 1 read + 1 write, bwd-reduce 2 reads, bwd-apply 2 reads + 1 write of M*C.
#include "common.h"

namespace {

// All four streaming kernels share one shape: block = 64 channel-quads x 4 row lanes, a thread keeps its quad's
// per-channel constants in registers and walks its rows four at a time (four independent 8/16-byte loads in flight).
constexpr int kRowUnroll = 4;
// Partial sums go to 32 workspace slots (block y % 32), not one: 1024 blocks adding into the same 2*C addresses
// serialised in the atomic unit (~100 us of a 133 us reduction); a fold kernel / the finalize kernel sums the slots.
constexpr int kBnSlots = CMDA_BN_SLOTS;

template <typename T>
__global__ void bn_reduce_kernel(const T* __restrict__ x, float* __restrict__ ws, long M, int C, int rows_per_block) {
  __shared__ float red[2][4][64][4];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cx) * 4;
  x += (long)blockIdx.z * M * C;                      // group z: rows [z*M, (z+1)*M), its own statistics
  ws += (long)blockIdx.z * (kBnSlots + 1) * 2 * C;
  const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (c < C) {
    float sh[4];
    ld4(x + c, sh);  // shift = row 0
    for (long r = r0 + ry; r < r1; r += 4 * kRowUnroll) {
      float v[kRowUnroll][4];
#pragma unroll
      for (int u = 0; u < kRowUnroll; ++u) {
        const long ru = r + 4 * u;
        ld4(x + (ru < r1 ? ru : r) * C + c, v[u]);
      }
#pragma unroll
      for (int u = 0; u < kRowUnroll; ++u) {
        if (r + 4 * u >= r1) break;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = v[u][j] - sh[j];
          s[j] += d;
          q[j] += d * d;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[0][ry][cx][j] = s[j];
    red[1][ry][cx][j] = q[j];
  }
  __syncthreads();
  if (ry == 0 && c < C) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* wsl = ws + (long)(blockIdx.y % kBnSlots) * 2 * C;
      atomicAdd(wsl + c + j, red[0][0][cx][j] + red[0][1][cx][j] + red[0][2][cx][j] + red[0][3][cx][j]);
      atomicAdd(wsl + C + c + j, red[1][0][cx][j] + red[1][1][cx][j] + red[1][2][cx][j] + red[1][3][cx][j]);
    }
  }
}

// 32 lanes per channel (one per workspace slot; 8 channels per 256-thread block): the slot sums of a group are one xor-shuffle
// reduction instead of a 64-load loop in one thread (24.8 -> ~6 us per launch, eight per head pass on its single lane); lane 0 applies the G
// groups' running-statistics updates one after the other in `order` (the shared decoder's BatchNorm sees image, events, fusion, ISR
// features in that order: daformer_head.py:305-319), which is what makes a grouped call equal to G separate calls.
struct BnOrder { int g[8]; };
template <typename T>
__global__ void bn_finalize_kernel(const T* __restrict__ x, const float* __restrict__ ws, float* __restrict__ mean,
                                   float* __restrict__ rstd, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, long M, int C, float eps, float momentum, int G,
                                   BnOrder order, int fused_stats) {
  static_assert(kBnSlots == 32, "one lane per slot");
  const int k = threadIdx.x & 31;
  const int c = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);
  const bool live = c < C;
  const int cc = live ? c : C - 1;   // (every lane takes part in the shuffles)
  float rm = 0.f, rv = 0.f;
  if (k == 0 && live && running_mean) {
    rm = running_mean[c];
    rv = running_var[c];
  }
  for (int i = 0; i < G; ++i) {
    const int g = order.g[i];
    float* wg = const_cast<float*>(ws) + (long)g * (kBnSlots + 1) * 2 * C;
    float s = wg[(long)k * 2 * C + cc], q = wg[(long)k * 2 * C + C + cc];
    if (fused_stats && live) {   // sums from a GEMM epilogue (cmda_gemm colstats): the workspace is handed back zeroed
      wg[(long)k * 2 * C + cc] = 0.f;
      wg[(long)k * 2 * C + C + cc] = 0.f;
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      s += __shfl_xor(s, o, 64);
      q += __shfl_xor(q, o, 64);
    }
    if (k != 0 || !live) continue;
    const float shift = fused_stats ? 0.f : ldf(x + (long)g * M * C + c);
    const float md = s / (float)M;
    const float mu = shift + md;
    float var = q / (float)M - md * md;
    var = fmaxf(var, 0.f);
    mean[(long)g * C + c] = mu;
    rstd[(long)g * C + c] = rsqrtf(var + eps);
    const float unb = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
    rm = (1.f - momentum) * rm + momentum * mu;
    rv = (1.f - momentum) * rv + momentum * unb;
  }
  if (k == 0 && live && running_mean) {
    running_mean[c] = rm;
    running_var[c] = rv;
  }
}

template <typename T>
__global__ void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y,
                                long M, int C, int relu, int ldy, int coff, int rows_per_block) {
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cx) * 4;
  if (c >= C) return;
  x += (long)blockIdx.z * M * C;
  y += (long)blockIdx.z * M * ldy;
  mean += (long)blockIdx.z * C;
  rstd += (long)blockIdx.z * C;
  const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float mu[4], rs[4], g[4], b[4];
  ld4(mean + c, mu);
  ld4(rstd + c, rs);
  ld4(gamma + c, g);
  ld4(beta + c, b);
  for (long r = r0 + ry; r < r1; r += 4 * kRowUnroll) {
    float v[kRowUnroll][4];
#pragma unroll
    for (int u = 0; u < kRowUnroll; ++u) {
      const long ru = r + 4 * u;
      ld4(x + (ru < r1 ? ru : r) * C + c, v[u]);
    }
#pragma unroll
    for (int u = 0; u < kRowUnroll; ++u) {
      const long ru = r + 4 * u;
      if (ru >= r1) break;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[u][j] = (v[u][j] - mu[j]) * rs[j] * g[j] + b[j];
        if (relu) v[u][j] = fmaxf(v[u][j], 0.f);
      }
      st4(y + ru * ldy + coff + c, v[u]);
    }
  }
}

// Mixed storage types (the Motion-Extractor generator in the bf16 mode): x in TX (the convolution's fp32 output -- its statistics and
// the normalisation see the unrounded sums), y in TY; optional fp32 residual added AFTER the normalisation (ResnetBlock:
// x + conv_block(x), cyclegan_model.py:431-434, the stream itself stays fp32) and an optional bf16 copy y2 of the result (pitch C),
// the operand of the next convolution.
template <typename TX, typename TY>
__global__ void bn_apply2_kernel(const TX* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                 const float* __restrict__ gamma, const float* __restrict__ beta, TY* __restrict__ y,
                                 const float* __restrict__ res, bf16_t* __restrict__ y2, long M, int C, int relu, int ldy,
                                 int rows_per_block) {
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cx) * 4;
  if (c >= C) return;
  x += (long)blockIdx.z * M * C;
  y += (long)blockIdx.z * M * ldy;
  if (res) res += (long)blockIdx.z * M * C;
  if (y2) y2 += (long)blockIdx.z * M * C;
  mean += (long)blockIdx.z * C;
  rstd += (long)blockIdx.z * C;
  const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float mu[4], rs[4], g[4], b[4];
  ld4(mean + c, mu);
  ld4(rstd + c, rs);
  ld4(gamma + c, g);
  ld4(beta + c, b);
  for (long r = r0 + ry; r < r1; r += 4 * kRowUnroll) {
    float v[kRowUnroll][4], q[kRowUnroll][4];
#pragma unroll
    for (int u = 0; u < kRowUnroll; ++u) {
      const long ru = r + 4 * u < r1 ? r + 4 * u : r;
      ld4(x + ru * C + c, v[u]);
      if (res) ld4(res + ru * C + c, q[u]);
    }
#pragma unroll
    for (int u = 0; u < kRowUnroll; ++u) {
      const long ru = r + 4 * u;
      if (ru >= r1) break;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[u][j] = (v[u][j] - mu[j]) * rs[j] * g[j] + b[j];
        if (relu) v[u][j] = fmaxf(v[u][j], 0.f);
        if (res) v[u][j] += q[u][j];
      }
      st4(y + ru * ldy + c, v[u]);
      if (y2) st4(y2 + ru * C + c, v[u]);
    }
  }
}

// ws[0:C] += sum dyr, ws[C:2C] += sum dyr*xhat, dyr = dy masked by the ReLU of the recomputed output
template <typename T>
__global__ void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ mean,
                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, float* __restrict__ ws, long M, int C, int relu,
                                     int lddy, int coff, int rows_per_block) {
  __shared__ float red[2][4][64][4];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cx) * 4;
  x += (long)blockIdx.z * M * C;
  dy += (long)blockIdx.z * M * lddy;
  mean += (long)blockIdx.z * C;
  rstd += (long)blockIdx.z * C;
  ws += (long)blockIdx.z * (kBnSlots + 1) * 2 * C;
  const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (c < C) {
    float mu[4], rs[4], g[4], b[4];
    ld4(mean + c, mu);
    ld4(rstd + c, rs);
    ld4(gamma + c, g);
    ld4(beta + c, b);
    for (long r = r0 + ry; r < r1; r += 4 * kRowUnroll) {
      float v[kRowUnroll][4], d[kRowUnroll][4];
#pragma unroll
      for (int u = 0; u < kRowUnroll; ++u) {
        const long ru = r + 4 * u < r1 ? r + 4 * u : r;
        ld4(x + ru * C + c, v[u]);
        ld4(dy + ru * lddy + coff + c, d[u]);
      }
#pragma unroll
      for (int u = 0; u < kRowUnroll; ++u) {
        if (r + 4 * u >= r1) break;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = (v[u][j] - mu[j]) * rs[j];
          float dj = d[u][j];
          if (relu && xh * g[j] + b[j] <= 0.f) dj = 0.f;
          s1[j] += dj;
          s2[j] += dj * xh;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[0][ry][cx][j] = s1[j];
    red[1][ry][cx][j] = s2[j];
  }
  __syncthreads();
  if (ry == 0 && c < C) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* wsl = ws + (long)(blockIdx.y % kBnSlots) * 2 * C;
      atomicAdd(wsl + c + j, red[0][0][cx][j] + red[0][1][cx][j] + red[0][2][cx][j] + red[0][3][cx][j]);
      atomicAdd(wsl + C + c + j, red[1][0][cx][j] + red[1][1][cx][j] + red[1][2][cx][j] + red[1][3][cx][j]);
    }
  }
}

// ws[slots][2C] -> ws[kBnSlots][2C] (the folded sums the apply pass reads)
__global__ void bn_fold_kernel(float* __restrict__ ws, int C2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C2) return;
  ws += (long)blockIdx.z * (kBnSlots + 1) * C2;
  float a = 0.f;
  for (int k = 0; k < kBnSlots; ++k) a += ws[(long)k * C2 + i];
  ws[(long)kBnSlots * C2 + i] = a;
}

// dx = gamma*rstd * (dyr - mean(dyr) - xhat * mean(dyr*xhat)); also accumulates dgamma / dbeta from ws
template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const float* __restrict__ ws, T* __restrict__ dx,
                                    float* __restrict__ dgamma, float* __restrict__ dbeta, long M, int C, int relu,
                                    int lddy, int coff, int rows_per_block) {
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cx) * 4;
  if (c >= C) return;
  x += (long)blockIdx.z * M * C;
  dx += (long)blockIdx.z * M * C;
  dy += (long)blockIdx.z * M * lddy;
  mean += (long)blockIdx.z * C;
  rstd += (long)blockIdx.z * C;
  ws += (long)blockIdx.z * (kBnSlots + 1) * 2 * C;
  const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  const float invM = 1.f / (float)M;
  float mu[4], rs[4], g[4], b[4], a1[4], a2[4];
  ld4(mean + c, mu);
  ld4(rstd + c, rs);
  ld4(gamma + c, g);
  ld4(beta + c, b);
  ld4(ws + c, a1);
  ld4(ws + C + c, a2);
  if (blockIdx.y == 0 && ry == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (gridDim.z > 1) {   // the groups share gamma / beta: their contributions meet in the same addresses
        atomicAdd(dbeta + c + j, a1[j]);
        atomicAdd(dgamma + c + j, a2[j]);
      } else {
        dbeta[c + j] += a1[j];
        dgamma[c + j] += a2[j];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    a1[j] *= invM;
    a2[j] *= invM;
  }
  for (long r = r0 + ry; r < r1; r += 4 * kRowUnroll) {
    float v[kRowUnroll][4], d[kRowUnroll][4];
#pragma unroll
    for (int u = 0; u < kRowUnroll; ++u) {
      const long ru = r + 4 * u < r1 ? r + 4 * u : r;
      ld4(x + ru * C + c, v[u]);
      ld4(dy + ru * lddy + coff + c, d[u]);
    }
#pragma unroll
    for (int u = 0; u < kRowUnroll; ++u) {
      const long ru = r + 4 * u;
      if (ru >= r1) break;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xh = (v[u][j] - mu[j]) * rs[j];
        float dj = d[u][j];
        if (relu && xh * g[j] + b[j] <= 0.f) dj = 0.f;
        v[u][j] = g[j] * rs[j] * (dj - a1[j] - xh * a2[j]);
      }
      st4(dx + ru * C + c, v[u]);
    }
  }
}

// reductions: few fat blocks (one fp32 atomic per channel per block); streaming applies: >= ~8 blocks per CU
static inline int rows_per_block(long M, int gx) {
  int rpb = 512;
  while (rpb > 32 && (M + rpb - 1) / rpb * gx < 1024) rpb >>= 1;
  return rpb;
}
static inline int rows_per_block_apply(long M, int gx) {
  int rpb = 256;
  while (rpb > 16 && (M + rpb - 1) / rpb * gx < 2048) rpb >>= 1;
  return rpb;
}
}  // namespace

// ws: cmda_bn_ws_floats(C) = 66*C floats of scratch PER GROUP (zeroed here).  Saves mean/rstd [groups][C] for the backward,
// updates running stats in place.  groups > 1: x / y hold `groups` consecutive blocks of M rows, each normalised with its own
// batch statistics (the shared decoder run once over the image / events / fusion / ISR features); `order` (HOST, may be
// NULL = 0,1,2..) lists the groups in the order their running-statistic updates are applied.
extern "C" int64_t cmda_bn_ws_floats(int C) { return (int64_t)(kBnSlots + 1) * 2 * C; }

extern "C" int cmda_bn_train_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                 float* running_mean, float* running_var, float* ws, int64_t M, int C, float eps,
                                 float momentum, int relu, int ldy, int coff, int groups, const int* order, int ws_has_stats,
                                 int dtype, void* stream) {
  if (M <= 0 || C <= 0) return CMDA_OK;
  if ((C & 3) || (ldy & 3) || (coff & 3) || groups < 1 || groups > 8) return CMDA_ERR_SHAPE;
  BnOrder ord;
  for (int i = 0; i < 8; ++i) ord.g[i] = (order && i < groups) ? order[i] : i;
  unsigned seen = 0;
  for (int i = 0; i < groups; ++i) {
    if (ord.g[i] < 0 || ord.g[i] >= groups) return CMDA_ERR_SHAPE;
    seen |= 1u << ord.g[i];
  }
  // with the statistics already in the workspace (ws_has_stats) the finalize pass is what hands it back ZEROED, group by group in
  // `order`: a repeated group would leave another group's sums in place for the next layer on this lane (advisor finding, round 5)
  if (ws_has_stats && seen != (1u << groups) - 1u) return CMDA_ERR_SHAPE;
  if (!ws_has_stats) cmda_zero_async(ws, sizeof(float) * (size_t)groups * (kBnSlots + 1) * 2 * C, stream);
  const int gx = (C / 4 + 63) / 64;
  const int rpb = rows_per_block(M * groups, gx);
  dim3 rgrid(gx, (unsigned)((M + rpb - 1) / rpb), groups);
  const int arpb = rows_per_block_apply(M * groups, gx);
  dim3 agrid(gx, (unsigned)((M + arpb - 1) / arpb), groups);
  CMDA_DISPATCH_DTYPE(dtype, {
    if (!ws_has_stats) CMDA_LAUNCH((bn_reduce_kernel<T>), rgrid, dim3(256), 0, stream, (const T*)x, ws, (long)M, C, rpb);
    CMDA_LAUNCH((bn_finalize_kernel<T>), dim3((C + 7) / 8), dim3(256), 0, stream, (const T*)x, ws, mean, rstd,
                running_mean, running_var, (long)M, C, eps, momentum, groups, ord, ws_has_stats);
    CMDA_LAUNCH((bn_apply_kernel<T>), agrid, dim3(256), 0, stream, (const T*)x, mean, rstd, gamma, beta, (T*)y + coff, (long)M,
                C, relu, ldy, 0, arpb);
  });
  CMDA_CHECK_LAUNCH();
}

// x_dtype / y_dtype independent; res32 (fp32 [groups*M, C]) is added after the normalisation (+ReLU); y2_bf16 (bf16 [groups*M, C])
// receives a second copy of the result.  Otherwise cmda_bn_train_fwd.
extern "C" int cmda_bn_train_fwd2(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype,
                                  float* mean, float* rstd, float* running_mean, float* running_var, float* ws, int64_t M, int C,
                                  float eps, float momentum, int relu, int ldy, int coff, int groups, const int* order,
                                  const float* res32, void* y2_bf16, int ws_has_stats, void* stream) {
  if (M <= 0 || C <= 0) return CMDA_OK;
  if ((C & 3) || (ldy & 3) || (coff & 3) || groups < 1 || groups > 8) return CMDA_ERR_SHAPE;
  if ((x_dtype != CMDA_F32 && x_dtype != CMDA_BF16) || (y_dtype != CMDA_F32 && y_dtype != CMDA_BF16)) return CMDA_ERR_DTYPE;
  BnOrder ord;
  for (int i = 0; i < 8; ++i) ord.g[i] = (order && i < groups) ? order[i] : i;
  unsigned seen = 0;
  for (int i = 0; i < groups; ++i) {
    if (ord.g[i] < 0 || ord.g[i] >= groups) return CMDA_ERR_SHAPE;
    seen |= 1u << ord.g[i];
  }
  // with the statistics already in the workspace (ws_has_stats) the finalize pass is what hands it back ZEROED, group by group in
  // `order`: a repeated group would leave another group's sums in place for the next layer on this lane (advisor finding, round 5)
  if (ws_has_stats && seen != (1u << groups) - 1u) return CMDA_ERR_SHAPE;
  if (!ws_has_stats) cmda_zero_async(ws, sizeof(float) * (size_t)groups * (kBnSlots + 1) * 2 * C, stream);
  const int gx = (C / 4 + 63) / 64;
  const int rpb = rows_per_block(M * groups, gx);
  dim3 rgrid(gx, (unsigned)((M + rpb - 1) / rpb), groups);
  const int arpb = rows_per_block_apply(M * groups, gx);
  dim3 agrid(gx, (unsigned)((M + arpb - 1) / arpb), groups);
#define CMDA_BN2_APPLY(TX, TY)                                                                                                    \
  CMDA_LAUNCH((bn_apply2_kernel<TX, TY>), agrid, dim3(256), 0, stream, (const TX*)x, mean, rstd, gamma, beta, (TY*)y + coff,      \
              res32, (bf16_t*)y2_bf16, (long)M, C, relu, ldy, arpb)
  CMDA_DISPATCH_DTYPE(x_dtype, {
    if (!ws_has_stats) CMDA_LAUNCH((bn_reduce_kernel<T>), rgrid, dim3(256), 0, stream, (const T*)x, ws, (long)M, C, rpb);
    CMDA_LAUNCH((bn_finalize_kernel<T>), dim3((C + 7) / 8), dim3(256), 0, stream, (const T*)x, ws, mean, rstd,
                running_mean, running_var, (long)M, C, eps, momentum, groups, ord, ws_has_stats);
    if (y_dtype == CMDA_F32) CMDA_BN2_APPLY(T, float); else CMDA_BN2_APPLY(T, bf16_t);
  });
#undef CMDA_BN2_APPLY
  CMDA_CHECK_LAUNCH();
}

// eval-mode / given-statistics apply (y = relu?((x-mean)*rstd*gamma+beta))
extern "C" int cmda_bn_apply(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                             void* y, int64_t M, int C, int relu, int ldy, int coff, int dtype, void* stream) {
  if (M <= 0 || C <= 0) return CMDA_OK;
  if ((C & 3) || (ldy & 3) || (coff & 3)) return CMDA_ERR_SHAPE;
  const int gx = (C / 4 + 63) / 64;
  const int arpb = rows_per_block_apply(M, gx);
  dim3 agrid(gx, (unsigned)((M + arpb - 1) / arpb));
  CMDA_DISPATCH_DTYPE(dtype, CMDA_LAUNCH((bn_apply_kernel<T>), agrid, dim3(256), 0, stream, (const T*)x, mean, rstd, gamma,
                                         beta, (T*)y, (long)M, C, relu, ldy, coff, arpb));
  CMDA_CHECK_LAUNCH();
}

// dy may be a channel slice [coff, coff+C) of a wider buffer with row pitch lddy; x is the pre-BN tensor [groups*M, C];
// mean / rstd are [groups][C] as saved by the forward call; dgamma / dbeta accumulate over all groups.
extern "C" int cmda_bn_train_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                                 const float* beta, void* dx, float* dgamma, float* dbeta, float* ws, int64_t M, int C,
                                 int relu, int lddy, int coff, int groups, int dtype, void* stream) {
  if (M <= 0 || C <= 0) return CMDA_OK;
  if ((C & 3) || (lddy & 3) || (coff & 3) || groups < 1 || groups > 8) return CMDA_ERR_SHAPE;
  cmda_zero_async(ws, sizeof(float) * (size_t)groups * (kBnSlots + 1) * 2 * C, stream);
  const int gx = (C / 4 + 63) / 64;
  const int rpb = rows_per_block(M * groups, gx);
  dim3 rgrid(gx, (unsigned)((M + rpb - 1) / rpb), groups);
  const int arpb = rows_per_block_apply(M * groups, gx);
  dim3 agrid(gx, (unsigned)((M + arpb - 1) / arpb), groups);
  CMDA_DISPATCH_DTYPE(dtype, {
    CMDA_LAUNCH((bn_bwd_reduce_kernel<T>), rgrid, dim3(256), 0, stream, (const T*)dy + coff, (const T*)x, mean, rstd, gamma,
                beta, ws, (long)M, C, relu, lddy, 0, rpb);
    CMDA_LAUNCH(bn_fold_kernel, dim3((2 * C + 255) / 256, 1, groups), dim3(256), 0, stream, ws, 2 * C);
    CMDA_LAUNCH((bn_bwd_apply_kernel<T>), agrid, dim3(256), 0, stream, (const T*)dy + coff, (const T*)x, mean, rstd, gamma, beta,
                ws + (long)kBnSlots * 2 * C, (T*)dx, dgamma, dbeta, (long)M, C, relu, lddy, 0, arpb);
  });
  CMDA_CHECK_LAUNCH();
}
