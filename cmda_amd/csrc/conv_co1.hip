// conv_co1.hip -- K x K convolution with ONE output channel (stride 1, reflection or zero padding) on NHWC bf16 activations: the last
// layer of the Image Motion-Extractor generator (ReflectionPad2d(3) + Conv2d(64, 1, 7) + Tanh, cyclegan_model.py:366-369).
//
// As a GEMM this is N = 1: the MFMA tile kernel spends a 64-wide tile on one column (510 us for 2 x 512 x 512 pixels).  Here a
// workgroup owns a 16 x 16 output tile: its (16 + K - 1)^2 input pixels go to LDS once (pixel pitch C + 8 bf16 = 144 bytes for
// C = 64, so the 16 lanes a ds_read_b128 services together -- 16 neighbouring pixels, same channels -- hit 16 different bank
// groups), the K*K*C weights beside them (read as broadcasts), and every thread accumulates its pixel's K*K*C products with
// v_dot2c_f32_bf16 (two bf16 products per lane per instruction, fp32 accumulate).  LDS-read bound: K*K*C*2 bytes per output.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

static __device__ __forceinline__ float dot8(const uint4& a, const uint4& b, float acc) {
#ifndef CMDA_EMU
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.x), __builtin_bit_cast(bf16x2_t, b.x), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.y), __builtin_bit_cast(bf16x2_t, b.y), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.z), __builtin_bit_cast(bf16x2_t, b.z), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.w), __builtin_bit_cast(bf16x2_t, b.w), acc, false);
  return acc;
#else
  const unsigned av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
  for (int i = 0; i < 4; ++i) {
    acc = fmaf(bf2f((bf16_t)(av[i] & 0xffff)), bf2f((bf16_t)(bv[i] & 0xffff)), acc);
    acc = fmaf(bf2f((bf16_t)(av[i] >> 16)), bf2f((bf16_t)(bv[i] >> 16)), acc);
  }
  return acc;
#endif
}

static __device__ __forceinline__ int reflect1(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

template <int K, int C>
__global__ __launch_bounds__(256) void conv_co1_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out, int H, int W,
                                                       int pad, int reflect, int act) {
  constexpr int T = 16, TW = T + K - 1, PITCH = C + 8, CH = C / 8;
  __shared__ __attribute__((aligned(16))) bf16_t sX[TW * TW * PITCH];
  __shared__ __attribute__((aligned(16))) bf16_t sW[K * K * C];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int ox0 = blockIdx.x * T, oy0 = blockIdx.y * T, b = blockIdx.z;
  const bf16_t* xb = x + (long)b * H * W * C;
  for (int i = tid; i < K * K * CH; i += 256) *reinterpret_cast<uint4*>(&sW[i * 8]) = *reinterpret_cast<const uint4*>(&w[i * 8]);
  for (int i = tid; i < TW * TW * CH; i += 256) {
    const int p = i / CH, c = i - p * CH;
    const int py = p / TW, px = p - py * TW;
    int ih = oy0 + py - pad, iw = ox0 + px - pad;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    bool ok = true;
    if (reflect) {
      ih = reflect1(ih, H);
      iw = reflect1(iw, W);
      ok = ih >= 0 && ih < H && iw >= 0 && iw < W;   // (tiles past the image edge: their outputs are never stored)
    } else {
      ok = ih >= 0 && ih < H && iw >= 0 && iw < W;
    }
    if (ok) v = *reinterpret_cast<const uint4*>(&xb[((long)ih * W + iw) * C + c * 8]);
    *reinterpret_cast<uint4*>(&sX[p * PITCH + c * 8]) = v;
  }
  __syncthreads();
  float acc0 = bias ? bias[0] : 0.f, acc1 = 0.f;
#pragma unroll 1
  for (int kh = 0; kh < K; ++kh) {
#pragma unroll
    for (int kw = 0; kw < K; ++kw) {
      const bf16_t* px = &sX[((ty + kh) * TW + tx + kw) * PITCH];
      const bf16_t* pw = &sW[(kh * K + kw) * C];
#pragma unroll
      for (int c = 0; c < CH; c += 2) {
        acc0 = dot8(*reinterpret_cast<const uint4*>(px + c * 8), *reinterpret_cast<const uint4*>(pw + c * 8), acc0);
        if (c + 1 < CH) acc1 = dot8(*reinterpret_cast<const uint4*>(px + c * 8 + 8), *reinterpret_cast<const uint4*>(pw + c * 8 + 8), acc1);
      }
    }
  }
  const int oy = oy0 + ty, ox = ox0 + tx;
  if (oy < H && ox < W) {
    float v = acc0 + acc1;
    if (act == 1) v = fmaxf(v, 0.f);
    else if (act == 3) v = tanhf(v);
    out[((long)b * H + oy) * W + ox] = v;
  }
}

// fp32 storage (the exact-fp32 and split-bf16 parity modes): plain FMAs, the channels in chunks of 16 so that the (16 + K - 1)^2
// pixel tile fits LDS (pixel pitch 20 floats = 80 bytes: the 16 pixels a ds_read_b128 services together start 20 banks apart).  As an
// N = 1 GEMM the layer cost the split-bf16 step 3.0 ms (524288 x 1 x 3136 at 1.1 TFLOP/s, round 5).
template <int K, int C>
__global__ __launch_bounds__(256) void conv_co1_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ out, int H, int W,
                                                           int pad, int reflect, int act) {
  constexpr int T = 16, TW = T + K - 1, CC = 16, PITCH = CC + 4, NCH = C / CC;
  __shared__ __attribute__((aligned(16))) float sX[TW * TW * PITCH];
  __shared__ __attribute__((aligned(16))) float sW[K * K * C];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int ox0 = blockIdx.x * T, oy0 = blockIdx.y * T, b = blockIdx.z;
  const float* xb = x + (long)b * H * W * C;
  for (int i = tid; i < K * K * C / 4; i += 256) *reinterpret_cast<float4*>(&sW[i * 4]) = *reinterpret_cast<const float4*>(&w[i * 4]);
  float acc[4] = {bias ? bias[0] : 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int cc = 0; cc < NCH; ++cc) {
    __syncthreads();   // (the previous chunk's tile has been consumed; first pass: nothing to wait for)
    for (int i = tid; i < TW * TW * (CC / 4); i += 256) {
      const int p = i / (CC / 4), c4 = i - p * (CC / 4);
      const int py = p / TW, px = p - py * TW;
      int ih = oy0 + py - pad, iw = ox0 + px - pad;
      if (reflect) {
        ih = reflect1(ih, H);
        iw = reflect1(iw, W);
      }
      const bool ok = ih >= 0 && ih < H && iw >= 0 && iw < W;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) v = *reinterpret_cast<const float4*>(&xb[((long)ih * W + iw) * C + cc * CC + c4 * 4]);
      *reinterpret_cast<float4*>(&sX[p * PITCH + c4 * 4]) = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int kh = 0; kh < K; ++kh) {
#pragma unroll
      for (int kw = 0; kw < K; ++kw) {
        const float* px = &sX[((ty + kh) * TW + tx + kw) * PITCH];
        const float* pw = &sW[(kh * K + kw) * C + cc * CC];
#pragma unroll
        for (int c4 = 0; c4 < CC / 4; ++c4) {
          const float4 a = *reinterpret_cast<const float4*>(px + c4 * 4), ww = *reinterpret_cast<const float4*>(pw + c4 * 4);
          acc[0] = fmaf(a.x, ww.x, acc[0]);
          acc[1] = fmaf(a.y, ww.y, acc[1]);
          acc[2] = fmaf(a.z, ww.z, acc[2]);
          acc[3] = fmaf(a.w, ww.w, acc[3]);
        }
      }
    }
  }
  const int oy = oy0 + ty, ox = ox0 + tx;
  if (oy < H && ox < W) {
    float v = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    if (act == 1) v = fmaxf(v, 0.f);
    else if (act == 3) v = tanhf(v);
    out[((long)b * H + oy) * W + ox] = v;
  }
}

}  // namespace

extern "C" int cmda_conv_co1(const void* x, const void* w, const float* bias, float* out, int B, int H, int W, int C, int K,
                             int pad, int reflect, int act, int dtype, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0) return CMDA_OK;
  if ((dtype != CMDA_BF16 && dtype != CMDA_F32) || K != 7 || C != 64 || pad != 3 || (act != 0 && act != 1 && act != 3)) return CMDA_ERR_UNSUPPORTED;
  if (reflect && (H <= pad || W <= pad)) return CMDA_ERR_SHAPE;
  const dim3 grid((W + 15) / 16, (H + 15) / 16, B);
  if (dtype == CMDA_F32) {
    CMDA_LAUNCH((conv_co1_f32_kernel<7, 64>), grid, dim3(256), 0, stream, (const float*)x, (const float*)w, bias, out, H, W, pad, reflect, act);
    CMDA_CHECK_LAUNCH();
  }
  CMDA_LAUNCH((conv_co1_kernel<7, 64>), grid, dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)w, bias, out, H, W, pad, reflect, act);
  CMDA_CHECK_LAUNCH();
}
