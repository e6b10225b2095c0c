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

template <typename T>
__global__ void bn_reduce_kernel(const T* __restrict__ x, float* __restrict__ ws, long M, int C, int rows_per_block) {
  __shared__ float red[2][4][64][4];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cx) * 4;
  const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (c < C) {
    float sh[4];
    ld4(x + c, sh);  // shift = row 0
    for (long r = r0 + ry; r < r1; r += 4) {
      float v[4];
      ld4(x + r * C + c, v);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = v[j] - sh[j];
        s[j] += d;
        q[j] += d * d;
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
      atomicAdd(ws + c + j, red[0][0][cx][j] + red[0][1][cx][j] + red[0][2][cx][j] + red[0][3][cx][j]);
      atomicAdd(ws + C + c + j, red[1][0][cx][j] + red[1][1][cx][j] + red[1][2][cx][j] + red[1][3][cx][j]);
    }
  }
}

template <typename T>
__global__ void bn_finalize_kernel(const T* __restrict__ x, const float* __restrict__ ws, float* __restrict__ mean,
                                   float* __restrict__ rstd, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, long M, int C, float eps, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float shift = ldf(x + c);
  const float s = ws[c], q = ws[C + c];
  const float md = s / (float)M;
  const float mu = shift + md;
  float var = q / (float)M - md * md;
  var = fmaxf(var, 0.f);
  mean[c] = mu;
  rstd[c] = rsqrtf(var + eps);
  if (running_mean) {
    const float unb = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
  }
}

template <typename T>
__global__ void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y,
                                long M, int C, int relu, int ldy, int coff) {
  const int cg = C >> 2;
  const long total = M * cg;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cg) * 4;
    const long r = i / cg;
    float v[4], mu[4], rs[4], g[4], b[4];
    ld4(x + r * C + c, v);
    ld4(mean + c, mu);
    ld4(rstd + c, rs);
    ld4(gamma + c, g);
    ld4(beta + c, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = (v[j] - mu[j]) * rs[j] * g[j] + b[j];
      if (relu) v[j] = fmaxf(v[j], 0.f);
    }
    st4(y + r * ldy + coff + c, v);
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
  const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (c < C) {
    float mu[4], rs[4], g[4], b[4];
    ld4(mean + c, mu);
    ld4(rstd + c, rs);
    ld4(gamma + c, g);
    ld4(beta + c, b);
    for (long r = r0 + ry; r < r1; r += 4) {
      float v[4], d[4];
      ld4(x + r * C + c, v);
      ld4(dy + r * lddy + coff + c, d);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xh = (v[j] - mu[j]) * rs[j];
        if (relu && xh * g[j] + b[j] <= 0.f) d[j] = 0.f;
        s1[j] += d[j];
        s2[j] += d[j] * xh;
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
      atomicAdd(ws + c + j, red[0][0][cx][j] + red[0][1][cx][j] + red[0][2][cx][j] + red[0][3][cx][j]);
      atomicAdd(ws + C + c + j, red[1][0][cx][j] + red[1][1][cx][j] + red[1][2][cx][j] + red[1][3][cx][j]);
    }
  }
}

// dx = gamma*rstd*(dyr - s1/M - xhat*s2/M); block 0 also folds s1/s2 into dbeta/dgamma
template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const float* __restrict__ ws, T* __restrict__ dx,
                                    float* __restrict__ dgamma, float* __restrict__ dbeta, long M, int C, int relu,
                                    int lddy, int coff) {
  const int cg = C >> 2;
  const long total = M * cg;
  const float invM = 1.f / (float)M;
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      dbeta[c] += ws[c];
      dgamma[c] += ws[C + c];
    }
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cg) * 4;
    const long r = i / cg;
    float v[4], d[4], mu[4], rs[4], g[4], b[4], a1[4], a2[4];
    ld4(x + r * C + c, v);
    ld4(dy + r * lddy + coff + c, d);
    ld4(mean + c, mu);
    ld4(rstd + c, rs);
    ld4(gamma + c, g);
    ld4(beta + c, b);
    ld4(ws + c, a1);
    ld4(ws + C + c, a2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (v[j] - mu[j]) * rs[j];
      if (relu && xh * g[j] + b[j] <= 0.f) d[j] = 0.f;
      v[j] = g[j] * rs[j] * (d[j] - a1[j] * invM - xh * a2[j] * invM);
    }
    st4(dx + r * C + c, v);
  }
}

static inline int grid_for(long n) { return (int)std::max<long>(1, std::min<long>((n + 255) / 256, 8192)); }
static inline int rows_per_block(long M, int gx) {
  int rpb = 512;
  while (rpb > 32 && (M + rpb - 1) / rpb * gx < 512) rpb >>= 1;
  return rpb;
}
}  // namespace

// ws: 2*C floats of scratch (zeroed here).  Saves mean/rstd [C] for the backward, updates running stats in place.
extern "C" int cmda_bn_train_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                 float* running_mean, float* running_var, float* ws, int64_t M, int C, float eps,
                                 float momentum, int relu, int ldy, int coff, int dtype, void* stream) {
  if (M <= 0 || C <= 0) return CMDA_OK;
  if ((C & 3) || (ldy & 3) || (coff & 3)) return CMDA_ERR_SHAPE;
  (void)hipMemsetAsync(ws, 0, sizeof(float) * 2 * C, (hipStream_t)stream);
  const int gx = (C / 4 + 63) / 64;
  const int rpb = rows_per_block(M, gx);
  dim3 rgrid(gx, (unsigned)((M + rpb - 1) / rpb));
  CMDA_DISPATCH_DTYPE(dtype, {
    CMDA_LAUNCH((bn_reduce_kernel<T>), rgrid, dim3(256), 0, stream, (const T*)x, ws, (long)M, C, rpb);
    CMDA_LAUNCH((bn_finalize_kernel<T>), dim3((C + 255) / 256), dim3(256), 0, stream, (const T*)x, ws, mean, rstd,
                running_mean, running_var, (long)M, C, eps, momentum);
    CMDA_LAUNCH((bn_apply_kernel<T>), dim3(grid_for(M * (C / 4))), dim3(256), 0, stream, (const T*)x, mean, rstd, gamma,
                beta, (T*)y, (long)M, C, relu, ldy, coff);
  });
  CMDA_CHECK_LAUNCH();
}

// eval-mode / given-statistics apply (y = relu?((x-mean)*rstd*gamma+beta))
extern "C" int cmda_bn_apply(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                             void* y, int64_t M, int C, int relu, int ldy, int coff, int dtype, void* stream) {
  if (M <= 0 || C <= 0) return CMDA_OK;
  if ((C & 3) || (ldy & 3) || (coff & 3)) return CMDA_ERR_SHAPE;
  CMDA_DISPATCH_DTYPE(dtype, CMDA_LAUNCH((bn_apply_kernel<T>), dim3(grid_for(M * (C / 4))), dim3(256), 0, stream,
                                         (const T*)x, mean, rstd, gamma, beta, (T*)y, (long)M, C, relu, ldy, coff));
  CMDA_CHECK_LAUNCH();
}

// dy may be a channel slice [coff, coff+C) of a wider buffer with row pitch lddy; x is the pre-BN tensor [M,C].
extern "C" int cmda_bn_train_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                                 const float* beta, void* dx, float* dgamma, float* dbeta, float* ws, int64_t M, int C,
                                 int relu, int lddy, int coff, int dtype, void* stream) {
  if (M <= 0 || C <= 0) return CMDA_OK;
  if ((C & 3) || (lddy & 3) || (coff & 3)) return CMDA_ERR_SHAPE;
  (void)hipMemsetAsync(ws, 0, sizeof(float) * 2 * C, (hipStream_t)stream);
  const int gx = (C / 4 + 63) / 64;
  const int rpb = rows_per_block(M, gx);
  dim3 rgrid(gx, (unsigned)((M + rpb - 1) / rpb));
  CMDA_DISPATCH_DTYPE(dtype, {
    CMDA_LAUNCH((bn_bwd_reduce_kernel<T>), rgrid, dim3(256), 0, stream, (const T*)dy, (const T*)x, mean, rstd, gamma,
                beta, ws, (long)M, C, relu, lddy, coff, rpb);
    CMDA_LAUNCH((bn_bwd_apply_kernel<T>), dim3(grid_for(M * (C / 4))), dim3(256), 0, stream, (const T*)dy, (const T*)x,
                mean, rstd, gamma, beta, ws, (T*)dx, dgamma, dbeta, (long)M, C, relu, lddy, coff);
  });
  CMDA_CHECK_LAUNCH();
}
