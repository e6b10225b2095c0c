// softmax.hip -- row softmax over attention scores (materialised-score attention path).
//
// Reference: `attn = (q @ k^T) * scale; attn = attn.softmax(dim=-1)` mix_transformer.py:97-98 and its autograd.
// One wave per row (row length L <= 1024: L = Nkv = 256 at 512x512, 260/280 at eval resolution), fp32 math,
// in-place.  HBM-bound: 2*rows*L*sizeof(T) bytes per pass.
#include "common.h"

namespace {
constexpr int kMaxPer = 16;  // 64 lanes * 16 = 1024

// P = softmax(alpha * S), in place
template <typename T>
__global__ void softmax_fwd_kernel(T* __restrict__ s, long rows, int L, float alpha) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  for (long row = (long)blockIdx.x * wpb + wid; row < rows; row += (long)gridDim.x * wpb) {
    T* sr = s + row * L;
    float v[kMaxPer];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < kMaxPer; ++i) {
      const int c = i * 64 + lane;
      v[i] = c < L ? alpha * ldf(sr + c) : -INFINITY;
      mx = fmaxf(mx, v[i]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxPer; ++i) {
      const int c = i * 64 + lane;
      v[i] = c < L ? __expf(v[i] - mx) : 0.f;
      sum += v[i];
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int i = 0; i < kMaxPer; ++i) {
      const int c = i * 64 + lane;
      if (c < L) stf(sr + c, v[i] * inv);
    }
  }
}

// dS = alpha * P * (dP - sum_j P_j dP_j), written over dP
template <typename T>
__global__ void softmax_bwd_kernel(const T* __restrict__ p, T* __restrict__ dp, long rows, int L, float alpha) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  for (long row = (long)blockIdx.x * wpb + wid; row < rows; row += (long)gridDim.x * wpb) {
    const T* pr = p + row * L;
    T* dr = dp + row * L;
    float pv[kMaxPer], dv[kMaxPer];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxPer; ++i) {
      const int c = i * 64 + lane;
      pv[i] = c < L ? ldf(pr + c) : 0.f;
      dv[i] = c < L ? ldf(dr + c) : 0.f;
      dot += pv[i] * dv[i];
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int i = 0; i < kMaxPer; ++i) {
      const int c = i * 64 + lane;
      if (c < L) stf(dr + c, alpha * pv[i] * (dv[i] - dot));
    }
  }
}
}  // namespace

extern "C" int cmda_softmax_fwd(void* s, int64_t rows, int L, float alpha, int dtype, void* stream) {
  if (rows <= 0) return CMDA_OK;
  if (L <= 0 || L > 64 * kMaxPer) return CMDA_ERR_SHAPE;
  const int grid = (int)std::min<long>((rows + 3) / 4, 8192);
  CMDA_DISPATCH_DTYPE(dtype, CMDA_LAUNCH((softmax_fwd_kernel<T>), dim3(grid), dim3(256), 0, stream, (T*)s, (long)rows, L, alpha));
  CMDA_CHECK_LAUNCH();
}

extern "C" int cmda_softmax_bwd(const void* p, void* dp, int64_t rows, int L, float alpha, int dtype, void* stream) {
  if (rows <= 0) return CMDA_OK;
  if (L <= 0 || L > 64 * kMaxPer) return CMDA_ERR_SHAPE;
  const int grid = (int)std::min<long>((rows + 3) / 4, 8192);
  CMDA_DISPATCH_DTYPE(dtype, CMDA_LAUNCH((softmax_bwd_kernel<T>), dim3(grid), dim3(256), 0, stream, (const T*)p, (T*)dp, (long)rows, L, alpha));
  CMDA_CHECK_LAUNCH();
}
