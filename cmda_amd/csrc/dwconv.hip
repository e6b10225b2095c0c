// dwconv.hip -- depthwise 3x3 convolution (optionally dilated) on NHWC activations.
//
// Reference ops replaced:
//   DWConv (3x3, pad 1, bias, groups=C) + GELU inside MixFFN   mix_transformer.py:37-44,443-455
//   depthwise half of DepthwiseSeparableConvModule (3x3, dilation 6/12/18, no bias) in the sep-ASPP
//   decode_heads/sep_aspp_head.py:18-27 (mmcv DepthwiseSeparableConvModule.depthwise_conv)
//
// HBM-bound stencils: a thread owns 4 adjacent channels of one pixel, lanes run along C so every tap is a
// coalesced 8/16-byte access; neighbouring pixels' taps hit L1/L2.  Algorithmic bytes per pixel-channel:
// fwd 2*sizeof(T); gelu-bwd-prep 3*sizeof(T); bwd-data 2*sizeof(T); bwd-weight 2*sizeof(T).
// Depthwise weights/bias stay fp32.  The stencil kernels read a tap-major [9][C] copy of the reference's [C,1,3,3]
// parameter (rt.wdw: one tiny permute per optimizer step) so that the 4 channel weights of a lane are one coalesced
// 16-byte load per tap; the weight-gradient kernel still accumulates in the parameter's own [C,9] layout.
#include "common.h"

namespace {

// Stencil kernels: block = 64 channel-quads x 4 pixel lanes; a thread keeps ITS 4 channels' nine taps (and bias) in
// registers and walks `pix_per_block / 4` pixels, so the per-pixel work is 9 coalesced 8/16-byte loads + FMAs.
// MODE 0: y = act(conv(x) + bias)          (forward)
// MODE 1: dz = da * gelu'(conv(x) + bias)  (backward prep: recomputes the pre-activation instead of saving it)
// MODE 2: dx (+)= conv^T(dy)               (data gradient: taps mirrored)
template <typename T, int MODE>
__global__ void dw_stencil_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                  const T* __restrict__ da, T* __restrict__ out, int B, int H, int W, int C, int dil, int act,
                                  int accumulate, int pix_per_block) {
  const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cx) * 4;
  if (c >= C) return;
  float wr[9][4], bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 9; ++t) ld4(w + t * C + c, wr[t]);
  if (MODE != 2 && bias) ld4(bias + c, bs);
  const long npix = (long)B * H * W;
  const long p0 = (long)blockIdx.y * pix_per_block;
  const long p1 = min(npix, p0 + pix_per_block);
  const int sgn = MODE == 2 ? -1 : 1;
  for (long pix = p0 + py; pix < p1; pix += 4) {
    const unsigned pu = (unsigned)pix;
    const int wx = (int)(pu % (unsigned)W);
    const unsigned t2 = pu / (unsigned)W;
    const int h = (int)(t2 % (unsigned)H);
    const int b = (int)(t2 / (unsigned)H);
    float acc[4] = {bs[0], bs[1], bs[2], bs[3]};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = h + sgn * (kh - 1) * dil;
      if (ih < 0 || ih >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = wx + sgn * (kw - 1) * dil;
        if (iw < 0 || iw >= W) continue;
        float xv[4];
        ld4(x + ((long)(b * H + ih) * W + iw) * C + c, xv);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += xv[j] * wr[kh * 3 + kw][j];
      }
    }
    T* o = out + pix * C + c;
    if (MODE == 0) {
      if (act == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = gelu_erf(acc[j]);
      }
    } else if (MODE == 1) {
      float g[4];
      ld4(da + pix * C + c, g);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = g[j] * gelu_erf_grad(acc[j]);
    } else if (accumulate) {
      float prev[4];
      ld4(o, prev);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += prev[j];
    }
    st4(o, acc);
  }
}

// dw[c,tap] += sum_pix dz[pix,c] * x[pix+tap,c];  dbias[c] += sum_pix dz[pix,c]
// block = 64 channel-groups x 4 pixel lanes; each block owns `pix_per_block` pixels; LDS reduce over the 4 pixel
// lanes, then one fp32 atomic per (channel, tap) per block.
template <typename T>
__global__ void dw_bwd_weight_kernel(const T* __restrict__ dz, const T* __restrict__ x, float* __restrict__ dw,
                                     float* __restrict__ dbias, int B, int H, int W, int C, int dil,
                                     int pix_per_block) {
  __shared__ float red[64][41];  // 10.5 KiB: the 4 pixel-lane waves fold into it with LDS float atomics
  const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
  for (int k = threadIdx.x; k < 64 * 41; k += blockDim.x) (&red[0][0])[k] = 0.f;
  __syncthreads();
  const int c = (blockIdx.x * 64 + cx) * 4;
  const long npix = (long)B * H * W;
  const long p0 = (long)blockIdx.y * pix_per_block;
  const long p1 = min(npix, p0 + pix_per_block);
  float acc[9][4], accb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = 0.f;
  if (c < C) {
    for (long pix = p0 + py; pix < p1; pix += 4) {
      const int wx = (int)(pix % W);
      const long t2 = pix / W;
      const int h = (int)(t2 % H);
      const int b = (int)(t2 / H);
      float g[4];
      ld4(dz + pix * C + c, g);
#pragma unroll
      for (int j = 0; j < 4; ++j) accb[j] += g[j];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int ih = h + (kh - 1) * dil;
        if (ih < 0 || ih >= H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int iw = wx + (kw - 1) * dil;
          if (iw < 0 || iw >= W) continue;
          float xv[4];
          ld4(x + ((long)(b * H + ih) * W + iw) * C + c, xv);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[kh * 3 + kw][j] += g[j] * xv[j];
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(&red[cx][t * 4 + j], acc[t][j]);
#pragma unroll
  for (int j = 0; j < 4; ++j) atomicAdd(&red[cx][36 + j], accb[j]);
  __syncthreads();
  // 64 channel groups x 40 values, summed over the 4 pixel lanes
  for (int k = threadIdx.x; k < 64 * 40; k += blockDim.x) {
    const int gx = k / 40, v = k - gx * 40;
    const int cc = (blockIdx.x * 64 + gx) * 4;
    if (cc >= C) continue;
    const float s = red[gx][v];
    if (v < 36) {
      const int t = v >> 2, j = v & 3;
      atomicAdd(dw + (cc + j) * 9 + t, s);
    } else if (dbias) {
      atomicAdd(dbias + cc + (v - 36), s);
    }
  }
}

static inline bool too_big(long n) { return n >= (1L << 32); }

template <int MODE>
static int launch_stencil(const void* x, const float* w, const float* bias, const void* da, void* out, int B, int H, int W,
                          int C, int dil, int act, int accumulate, int dtype, void* stream) {
  const long npix = (long)B * H * W;
  if (npix * C <= 0) return CMDA_OK;
  if ((C & 3) || too_big(npix)) return CMDA_ERR_SHAPE;
  const int gx = (C / 4 + 63) / 64;
  int ppb = 64;
  while (ppb > 8 && (npix + ppb - 1) / ppb * gx < 4096) ppb >>= 1;
  dim3 grid(gx, (unsigned)((npix + ppb - 1) / ppb));
  CMDA_DISPATCH_DTYPE(dtype, CMDA_LAUNCH((dw_stencil_kernel<T, MODE>), grid, dim3(256), 0, stream, (const T*)x, w, bias,
                                         (const T*)da, (T*)out, B, H, W, C, dil, act, accumulate, ppb));
  CMDA_CHECK_LAUNCH();
}
}  // namespace

extern "C" int cmda_dwconv3x3_fwd(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C,
                                  int dil, int act, int dtype, void* stream) {
  return launch_stencil<0>(x, w, bias, nullptr, y, B, H, W, C, dil, act, 0, dtype, stream);
}

extern "C" int cmda_dwconv3x3_gelu_bwd_prep(const void* x, const float* w, const float* bias, const void* da, void* dz,
                                            int B, int H, int W, int C, int dil, int dtype, void* stream) {
  return launch_stencil<1>(x, w, bias, da, dz, B, H, W, C, dil, 2, 0, dtype, stream);
}

extern "C" int cmda_dwconv3x3_bwd_data(const void* dy, const float* w, void* dx, int B, int H, int W, int C, int dil,
                                       int accumulate, int dtype, void* stream) {
  return launch_stencil<2>(dy, w, nullptr, nullptr, dx, B, H, W, C, dil, 0, accumulate, dtype, stream);
}

extern "C" int cmda_dwconv3x3_bwd_weight(const void* dz, const void* x, float* dw, float* dbias, int B, int H, int W,
                                         int C, int dil, int dtype, void* stream) {
  const long npix = (long)B * H * W;
  if (npix * C <= 0) return CMDA_OK;
  if (C & 3) return CMDA_ERR_SHAPE;
  const int gx = (C / 4 + 63) / 64;
  int ppb = 512;
  while (ppb > 32 && (npix + ppb - 1) / ppb * gx < 4096) ppb >>= 1;  // short per-thread pixel loops: latency-bound otherwise
  dim3 grid(gx, (unsigned)((npix + ppb - 1) / ppb));
  CMDA_DISPATCH_DTYPE(dtype, CMDA_LAUNCH((dw_bwd_weight_kernel<T>), grid, dim3(256), 0, stream, (const T*)dz,
                                         (const T*)x, dw, dbias, B, H, W, C, dil, ppb));
  CMDA_CHECK_LAUNCH();
}
