// elementwise.hip -- HBM-bound data movement / pointwise kernels.
//
//  cmda_permute4     : 4-D permute + cast (+flip, +accumulate).  Replaces the NLC<->NCHW `.permute().contiguous()`
//                      copies of mix_transformer.py:406,414,422,430 at the module boundary only, and repacks conv
//                      weights between the reference's [Co,Ci,KH,KW] parameter layout and the GEMM layouts.
//  cmda_colsum       : bias gradient (column sum of dY), fp32 accumulate.
//  cmda_axpby        : out = a*x + b*y  (feature averaging, attention_avg_fusion.py:49).
//  cmda_ema_update   : EMA teacher update, dacs.py:261-272.
//  cmda_adamw_step   : fused AdamW over a flat parameter segment (torch.optim.AdamW semantics,
//                      configs/_base_/schedules/adamw.py), optionally emitting the bf16 compute copy.
//  cmda_class_mix    : ClassMix of image / events / label / weight, dacs_transforms.py:101-131, dacs.py:716-771.
#include <cstdlib>
#include "common.h"

namespace {

template <typename TS, typename TD>
__global__ void permute4_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int d0, int d1, int d2, int d3,
                                int p0, int p1, int p2, int p3, int flipmask, int accumulate) {
  const int sd[4] = {d0, d1, d2, d3};
  const long ss[4] = {(long)d1 * d2 * d3, (long)d2 * d3, (long)d3, 1};
  const int pp[4] = {p0, p1, p2, p3};
  const long total = (long)d0 * d1 * d2 * d3;
  const int e1 = sd[pp[1]], e2 = sd[pp[2]], e3 = sd[pp[3]];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long t = i;
    int idx[4];
    idx[3] = (int)(t % e3); t /= e3;
    idx[2] = (int)(t % e2); t /= e2;
    idx[1] = (int)(t % e1); t /= e1;
    idx[0] = (int)t;
    long so = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      int v = idx[a];
      const int ax = pp[a];
      if ((flipmask >> ax) & 1) v = sd[ax] - 1 - v;
      so += (long)v * ss[ax];
    }
    float v = ldf(src + so);
    if (accumulate) v += ldf(dst + i);
    stf(dst + i, v);
  }
}

// dst = cast(src); src = 0 -- drains a persistent fp32 accumulation workspace (attention dK|dV: atomics from many query blocks)
// and leaves it zeroed for its next user, so neither a memset nor a separate cast launch is needed per call
template <typename TD>
__global__ void cast_clear_kernel(float* __restrict__ src, TD* __restrict__ dst, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float v[4];
    ld4(src + i * 4, v);
    st4(dst + i * 4, v);
    const float z[4] = {0.f, 0.f, 0.f, 0.f};
    st4(src + i * 4, z);
  }
}

// Batched form: one launch re-lays-out MANY tensors (every conv / depthwise weight of the student and the teacher after an
// optimizer / EMA step: ~470 tiny launches otherwise, and the 2 + 2 step is bound by launch count).  desc[t] describes tensor t;
// blocks[b] = {tensor, first element} gives every thread block its 1024-element chunk.  Sources are fp32 masters.
struct PermuteDesc {
  const float* src;
  void* dst;
  int d[4];
  int p[4];
  int flipmask;
  int dst_bf16;
  long total;
};

__global__ void permute4_batch_kernel(const PermuteDesc* __restrict__ desc, const int* __restrict__ blocks) {
  const int bk_x = blocks[2 * blockIdx.x], bk_y = blocks[2 * blockIdx.x + 1];
  const PermuteDesc D = desc[bk_x];
  // channel padding (flipmask bits 8-10 = padded source axis + 1, bits 16.. = its REAL extent): d[] are the padded dims the
  // destination is laid out with, the source holds only `pad_real` entries along that axis, the rest of the destination is zero
  // (the 7x7 patch-embed convolution's 3 input channels padded to 8: its im2col rows become 16-byte chunks of the LDS-DMA GEMM)
  const int pad_ax = ((D.flipmask >> 8) & 7) - 1, pad_real = D.flipmask >> 16;
  int rd[4] = {D.d[0], D.d[1], D.d[2], D.d[3]};
  if (pad_ax >= 0) rd[pad_ax] = pad_real;
  const long ss[4] = {(long)rd[1] * rd[2] * rd[3], (long)rd[2] * rd[3], (long)rd[3], 1};
  const int e1 = D.d[D.p[1]], e2 = D.d[D.p[2]], e3 = D.d[D.p[3]];
  const long base = (long)bk_y * 1024;
  if (D.dst_bf16 == 2) {
    // drain mode (conv weight gradients accumulated in the GEMM's own [Co][KH][KW][Ci] order): dst (fp32, [Co][Ci][KH][KW]) += src,
    // src = 0 -- the shadow is left zeroed for the next backward pass.  The chunk is walked in DST order, 4 consecutive elements
    // per thread: the read-modify-write of the gradient is one 16-byte access per lane (fully coalesced), the shadow is read
    // (and cleared) at (cell, ci) with ci advancing along the lanes -- whole sectors per cell row.  (Walking the source in order
    // made every gradient access a 4-byte scatter at stride KH*KW*4 bytes: 0.7 ms per step.)
    const int CiS = D.d[3], cells = D.d[1] * D.d[2];   // source dims (Co, KH, KW, Ci), p = (0, 3, 1, 2)
    const int Ci = pad_ax == 3 ? pad_real : CiS;       // (padded shadow: only the real channels exist in the gradient)
    float* src = const_cast<float*>(D.src);
    float* dst = reinterpret_cast<float*>(D.dst);
    const long i0 = base + 4L * threadIdx.x;
    if (i0 >= D.total) return;
    float add[4];
    bool any = false;
    const int n = (int)min(4L, D.total - i0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      add[e] = 0.f;
      if (e < n) {
        const long i = i0 + e;
        const long pair = i / cells;                 // co * Ci + ci
        const int cell = (int)(i - pair * cells);
        const long co = pair / Ci;
        const int ci = (int)(pair - co * Ci);
        float* sp = src + (co * cells + cell) * CiS + ci;
        add[e] = *sp;
        if (add[e] != 0.f) { *sp = 0.f; any = true; }
      }
    }
    if (!any) return;
    if (n == 4 && (i0 & 3) == 0) {
      float4 g = *reinterpret_cast<float4*>(dst + i0);
      g.x += add[0]; g.y += add[1]; g.z += add[2]; g.w += add[3];
      *reinterpret_cast<float4*>(dst + i0) = g;
    } else {
      for (int e = 0; e < n; ++e) dst[i0 + e] += add[e];
    }
    return;
  }
  if (D.p[0] == 0 && D.p[1] == 2 && D.p[2] == 3 && D.p[3] == 1 && D.flipmask == 0 && (D.total & 3) == 0 &&
      (reinterpret_cast<uintptr_t>(D.src) & 15) == 0) {
    // [Co][Ci][KH][KW] -> [Co][KH][KW][Ci] (the implicit-GEMM weight layout: most of the bytes of a refresh): walk the SOURCE in order,
    // one 16-byte load per lane; the stores of one (kh,kw) cell then run along ci across the lanes (whole sectors) instead of every
    // load being a 4-byte gather at stride KH*KW*4 bytes
    const int Ci = D.d[1], cells = D.d[2] * D.d[3];
    const long i0 = base + 4L * threadIdx.x;
    if (i0 >= D.total) return;
    float v[4];
    ld4(D.src + i0, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long i = i0 + e;
      const long pair = i / cells;                  // co * Ci + ci
      const int cell = (int)(i - pair * cells);
      const long co = pair / Ci;
      const int ci = (int)(pair - co * Ci);
      const long o = (co * cells + cell) * Ci + ci;
      if (D.dst_bf16) stf(reinterpret_cast<bf16_t*>(D.dst) + o, v[e]);
      else reinterpret_cast<float*>(D.dst)[o] = v[e];
    }
    return;
  }
  for (int k = threadIdx.x; k < 1024; k += blockDim.x) {
    const long i = base + k;
    if (i >= D.total) break;
    long t = i;
    int idx[4];
    idx[3] = (int)(t % e3); t /= e3;
    idx[2] = (int)(t % e2); t /= e2;
    idx[1] = (int)(t % e1); t /= e1;
    idx[0] = (int)t;
    long so = 0;
    bool in_pad = false;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      int v = idx[a];
      const int ax = D.p[a];
      if (ax == pad_ax && v >= pad_real) in_pad = true;
      if ((D.flipmask >> ax) & 1) v = rd[ax] - 1 - v;
      so += (long)v * ss[ax];
    }
    const float v = in_pad ? 0.f : D.src[so];
    if (D.dst_bf16) stf(reinterpret_cast<bf16_t*>(D.dst) + i, v);
    else reinterpret_cast<float*>(D.dst)[i] = v;
  }
}

// NCHW fp32 image -> NHWC rows with the channels padded to `cpad` (zeros): the encoder's input layout.  One thread per pixel: the C
// loads of a lane group run along W (coalesced per channel), the store is one contiguous cpad-element row.
template <typename TD>
__global__ void nchw_to_nhwc_pad_kernel(const float* __restrict__ src, TD* __restrict__ dst, int B, int C, long HW, int cpad) {
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW, pix = i - b * HW;
    const float* s = src + b * C * HW + pix;
    TD* d = dst + i * cpad;
    for (int c = 0; c < cpad; ++c) stf(d + c, c < C ? s[(long)c * HW] : 0.f);
  }
}

// identity permutation = a dtype cast (the flat fp32 -> bf16 parameter mirrors: up to 178 M elements per call): 4 elements per
// thread, no index arithmetic
template <typename TS, typename TD>
__global__ void cast4_kernel(const TS* __restrict__ src, TD* __restrict__ dst, long n4, int accumulate) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float v[4];
    ld4(src + i * 4, v);
    if (accumulate) {
      float o[4];
      ld4(dst + i * 4, o);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += o[j];
    }
    st4(dst + i * 4, v);
  }
}

// column sum of x[M,N] (row pitch ld) accumulated into out[N].  Block = 32 column-quads x 8 row lanes: lanes read 4
// adjacent columns (8/16-byte accesses) of 8 different rows; LDS reduce over the row lanes, one fp32 atomic per column
// per block.
template <typename T>
__global__ void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, long M, int N, long ld, int rows_per_block,
                              int vec) {
  __shared__ float red[8][32][4];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int col = (blockIdx.x * 32 + cx) * 4;
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = min(M, r0 + rows_per_block);
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (vec && col + 4 <= N) {
    for (long r = r0 + ry; r < r1; r += 8) {
      float v[4];
      ld4(x + r * ld + col, v);
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] += v[j];
    }
  } else if (col < N) {
    for (long r = r0 + ry; r < r1; r += 8)
      for (int j = 0; j < 4 && col + j < N; ++j) s[j] += ldf(x + r * ld + col + j);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[ry][cx][j] = s[j];
  __syncthreads();
  if (ry == 0 && col < N) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (col + j >= N) break;
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += red[k][cx][j];
      atomicAdd(out + col + j, t);
    }
  }
}

template <typename T>
__global__ void axpby_kernel(const T* __restrict__ x, const T* __restrict__ y, T* __restrict__ out, float a, float b,
                             long n) {
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
    if (i + 4 <= n) {
      float xv[4], yv[4], o[4];
      ld4(x + i, xv);
      if (y) ld4(y + i, yv); else yv[0] = yv[1] = yv[2] = yv[3] = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = a * xv[j] + b * yv[j];
      st4(out + i, o);
    } else {
      for (long j = i; j < n; ++j) stf(out + j, a * ldf(x + j) + (y ? b * ldf(y + j) : 0.f));
    }
  }
}

// out[b, i, c] = x[b, i, c] * scale[b * sstride + (c if per_channel)]  (DropPath per-sample scale, Dropout2d per
// (sample, channel) mask; timm DropPath mix_transformer.py:134,145-146, nn.Dropout2d decode_head.py:565-566)
template <typename T>
__global__ void sample_scale_kernel(const T* __restrict__ x, const float* __restrict__ scale, T* __restrict__ out,
                                    long per_sample, int C, int per_channel, long n) {
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
    // (32-bit index arithmetic when the tensor allows it: 64-bit integer division is emulated, ~70 instructions)
    long b;
    int c;
    if (n < (1L << 32)) {
      const unsigned iu = (unsigned)i, bu = iu / (unsigned)per_sample;
      b = bu;
      c = (int)(iu % (unsigned)C);
    } else {
      b = i / per_sample;
      c = (int)(i % C);
    }
    float v[4];
    ld4(x + i, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] *= per_channel ? scale[b * C + c + j] : scale[b];
    st4(out + i, v);
  }
}

// strided 2-D copy (channel concat / split of NLC tensors: attention_fusion.py:52, torch.cat)
template <typename T>
__global__ void copy2d_kernel(const T* __restrict__ src, T* __restrict__ dst, long rows, int cols, long src_ld,
                              long dst_ld) {
  const int cg = cols >> 2;
  const long total = rows * cg;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r;
    if (total < (1L << 32)) r = (unsigned)i / (unsigned)cg;   // (32-bit division: see sample_scale_kernel)
    else r = i / cg;
    const int c = (int)(i - r * cg) * 4;
    float v[4];
    ld4(src + r * src_ld + c, v);
    st4(dst + r * dst_ld + c, v);
  }
}

// Non-temporal 16-byte accesses for the once-per-step streams over the parameter state (AdamW: 30 bytes per parameter, EMA: 14).
#ifndef CMDA_EMU
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4_nt __attribute__((ext_vector_type(4)));
template <bool NT> static __device__ __forceinline__ void ld4s(const float* p, float (&v)[4]) {
  if constexpr (NT) {
    const f32x4_nt t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(p));
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  } else ld4(p, v);
}
template <bool NT> static __device__ __forceinline__ void st4s(float* p, const float (&v)[4]) {
  if constexpr (NT) {
    const f32x4_nt t = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4_nt*>(p));
  } else st4(p, v);
}
template <bool NT> static __device__ __forceinline__ void st4s(bf16_t* p, const float (&v)[4]) {
  if constexpr (NT) {
    const u16x4_nt t = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
    __builtin_nontemporal_store(t, reinterpret_cast<u16x4_nt*>(p));
  } else st4(p, v);
}
#else
template <bool NT> static inline void ld4s(const float* p, float (&v)[4]) { ld4(p, v); }
template <bool NT> static inline void st4s(float* p, const float (&v)[4]) { st4(p, v); }
template <bool NT> static inline void st4s(bf16_t* p, const float (&v)[4]) { st4(p, v); }
#endif


template <bool NT>
__global__ void ema_kernel(float* __restrict__ ema, const float* __restrict__ p, float alpha, long n, bf16_t* __restrict__ mirror) {
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (long i0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i0 < n; i0 += 2 * stride) {
    float e[2][4], v[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {   // both vectors requested before either is used
      const long i = i0 + u * stride;
      if (i + 4 <= n) {
        ld4s<NT>(ema + i, e[u]);
        ld4s<NT>(p + i, v[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long i = i0 + u * stride;
      if (i + 4 <= n) {
#pragma unroll
        for (int j = 0; j < 4; ++j) e[u][j] = alpha * e[u][j] + (1.f - alpha) * v[u][j];
        st4s<NT>(ema + i, e[u]);
        if (mirror) st4s<NT>(mirror + i, e[u]);   // the teacher's bf16 compute copy, written in the same pass (was a separate cast launch)
      } else if (i < n) {
        for (long j = i; j < n; ++j) {
          ema[j] = alpha * ema[j] + (1.f - alpha) * p[j];
          if (mirror) stf(mirror + j, ema[j]);
        }
      }
    }
  }
}

// torch.optim.AdamW (amsgrad=False): p *= 1 - lr*wd; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
static __device__ __forceinline__ float adamw_one(float pv, float gv, float& mv, float& vv, float lr, float b1, float b2, float eps,
                                                  float wd, float bc1, float bc2_sqrt) {
  pv *= 1.f - lr * wd;
  mv = b1 * mv + (1.f - b1) * gv;
  vv = b2 * vv + (1.f - b2) * gv * gv;
  const float denom = sqrtf(vv) / bc2_sqrt + eps;
  return pv - (lr / bc1) * (mv / denom);
}
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, bf16_t* __restrict__ p_bf16, long n, float lr, float b1, float b2,
                             float eps, float wd, float bc1, float bc2_sqrt) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float mv = m[i], vv = v[i];
    const float pv = adamw_one(p[i], g[i], mv, vv, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
    m[i] = mv;
    v[i] = vv;
    p[i] = pv;
    if (p_bf16) p_bf16[i] = f2bf(pv);
  }
}
// the same update on 16-byte vectors, two per thread and pass (8 independent 16-byte loads in flight per thread: the scalar kernel
// moved 30 bytes per parameter at ~2 TB/s); element-wise identical arithmetic.  n4 = vectors; the caller runs the scalar kernel
// over the ragged tail / unaligned groups.
template <bool NT>
__global__ void adamw_vec_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, bf16_t* __restrict__ p_bf16, long n4, float lr, float b1, float b2,
                                 float eps, float wd, float bc1, float bc2_sqrt) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += 2 * stride) {
    float pv[2][4], gv[2][4], mv[2][4], vv[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long i = i0 + u * stride;
      if (i < n4) {
        ld4s<NT>(p + 4 * i, pv[u]);
        ld4s<NT>(g + 4 * i, gv[u]);
        ld4s<NT>(m + 4 * i, mv[u]);
        ld4s<NT>(v + 4 * i, vv[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long i = i0 + u * stride;
      if (i < n4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) pv[u][j] = adamw_one(pv[u][j], gv[u][j], mv[u][j], vv[u][j], lr, b1, b2, eps, wd, bc1, bc2_sqrt);
        st4s<NT>(m + 4 * i, mv[u]);
        st4s<NT>(v + 4 * i, vv[u]);
        st4s<NT>(p + 4 * i, pv[u]);
        if (p_bf16) st4s<NT>(p_bf16 + 4 * i, pv[u]);
      }
    }
  }
}

// ClassMix for one tensor pair: out = m*src + (1-m)*tgt, mask m[b,h,w] = 1 where the source label is one of the
// chosen classes (dacs_transforms.py:121-131).  `classes` holds the chosen class ids per sample, padded with -1.
template <typename T>
__global__ void class_mix_kernel(const T* __restrict__ src, const T* __restrict__ tgt, T* __restrict__ out,
                                 const long long* __restrict__ src_label, const long long* __restrict__ classes,
                                 int max_classes, int B, int HW, int Cch, int channels_last) {
  const long total = (long)B * HW * Cch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long b, pix;
    if (total < (1L << 32)) {   // (32-bit division: see sample_scale_kernel)
      const unsigned iu = (unsigned)i;
      if (channels_last) {
        const unsigned bp = iu / (unsigned)Cch, bu = bp / (unsigned)HW;
        b = bu;
        pix = bp - bu * (unsigned)HW;
      } else {
        b = iu / ((unsigned)Cch * (unsigned)HW);
        pix = iu % (unsigned)HW;
      }
    } else if (channels_last) {
      const long bp = i / Cch;
      b = bp / HW;
      pix = bp - b * HW;
    } else {
      b = i / ((long)Cch * HW);
      pix = i % HW;
    }
    const long long lab = src_label[b * HW + pix];
    int msk = 0;
    for (int k = 0; k < max_classes; ++k) msk += (classes[b * max_classes + k] == lab) ? 1 : 0;
    const float mf = (float)msk;
    stf(out + i, mf * ldf(src + i) + (1.f - mf) * ldf(tgt + i));
  }
}

__global__ void class_mix_label_kernel(const long long* __restrict__ src, const long long* __restrict__ tgt,
                                       long long* __restrict__ out, const long long* __restrict__ src_label,
                                       const long long* __restrict__ classes, int max_classes, int B, int HW) {
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = total < (1L << 32) ? (long)((unsigned)i / (unsigned)HW) : i / HW;
    const long long lab = src_label[i];
    long long msk = 0;
    for (int k = 0; k < max_classes; ++k) msk += (classes[b * max_classes + k] == lab) ? 1 : 0;
    out[i] = msk * src[i] + (1 - msk) * tgt[i];
  }
}

static inline int grid_for(long n, int per_thread = 1) {
  long blocks = (n + 256L * per_thread - 1) / (256L * per_thread);
  return (int)std::max<long>(1, std::min<long>(blocks, 2048));
}

template <typename TS, typename TD>
int launch_permute(const void* src, void* dst, const int* d, const int* p, int flipmask, int accumulate, void* stream) {
  const long total = (long)d[0] * d[1] * d[2] * d[3];
  const bool identity = p[0] == 0 && p[1] == 1 && p[2] == 2 && p[3] == 3 && flipmask == 0;
  if (identity && (total & 3) == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0) {
    CMDA_LAUNCH((cast4_kernel<TS, TD>), dim3(grid_for(total / 4)), dim3(256), 0, stream, (const TS*)src, (TD*)dst, total / 4,
                accumulate);
    CMDA_CHECK_LAUNCH();
  }
  CMDA_LAUNCH((permute4_kernel<TS, TD>), dim3(grid_for(total)), dim3(256), 0, stream, (const TS*)src, (TD*)dst, d[0],
              d[1], d[2], d[3], p[0], p[1], p[2], p[3], flipmask, accumulate);
  CMDA_CHECK_LAUNCH();
}

}  // namespace

// dst dims = (d[p0], d[p1], d[p2], d[p3]); dst[i0..i3] = src[...] with src axis p_a indexed by i_a
// (reversed when bit p_a of flipmask is set); accumulate: dst += instead of dst =.
extern "C" int cmda_permute4(const void* src, void* dst, int d0, int d1, int d2, int d3, int p0, int p1, int p2, int p3,
                             int flipmask, int accumulate, int src_dtype, int dst_dtype, void* stream) {
  const int d[4] = {d0, d1, d2, d3}, p[4] = {p0, p1, p2, p3};
  if ((long)d0 * d1 * d2 * d3 <= 0) return CMDA_OK;
  int seen = 0;
  for (int a = 0; a < 4; ++a) {
    if (p[a] < 0 || p[a] > 3) return CMDA_ERR_SHAPE;
    seen |= 1 << p[a];
  }
  if (seen != 15) return CMDA_ERR_SHAPE;
  if (flipmask & ~15) return CMDA_ERR_UNSUPPORTED;   // channel-padding bits (8.., 16..) are cmda_permute4_batch's: not understood here
  if (src_dtype == CMDA_F32 && dst_dtype == CMDA_F32) return launch_permute<float, float>(src, dst, d, p, flipmask, accumulate, stream);
  if (src_dtype == CMDA_F32 && dst_dtype == CMDA_BF16) return launch_permute<float, bf16_t>(src, dst, d, p, flipmask, accumulate, stream);
  if (src_dtype == CMDA_BF16 && dst_dtype == CMDA_F32) return launch_permute<bf16_t, float>(src, dst, d, p, flipmask, accumulate, stream);
  if (src_dtype == CMDA_BF16 && dst_dtype == CMDA_BF16) return launch_permute<bf16_t, bf16_t>(src, dst, d, p, flipmask, accumulate, stream);
  return CMDA_ERR_DTYPE;
}

// hi = bf16(x), lo = bf16(x - hi): the split-bf16 operand pair of a large GEMM in the tolerance-meeting mode, written ONCE to HBM so
// that the bf16 LDS-DMA kernels can run the three products a_lo b_hi + a_hi b_lo + a_hi b_hi (8 bytes of traffic per element)
namespace {
__global__ void split_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float v[4], h[4], l[4];
    ld4(src + i * 4, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = bf2f(f2bf(v[e]));
      l[e] = v[e] - h[e];
    }
    st4(hi + i * 4, h);
    st4(lo + i * 4, l);
  }
}
}  // namespace

extern "C" int cmda_split_bf16(const float* src, void* hi, void* lo, int64_t n, void* stream) {
  if (n <= 0) return CMDA_OK;
  if ((n & 3) || ((uintptr_t)src % 16) || ((uintptr_t)hi % 8) || ((uintptr_t)lo % 8)) return CMDA_ERR_SHAPE;
  CMDA_LAUNCH(split_bf16_kernel, dim3(grid_for(n, 4)), dim3(256), 0, stream, src, (bf16_t*)hi, (bf16_t*)lo, (long)(n / 4));
  CMDA_CHECK_LAUNCH();
}

extern "C" int cmda_cast_clear(float* src, void* dst, int64_t n, int dst_dtype, void* stream) {
  if (n <= 0) return CMDA_OK;
  if ((n & 3) || ((uintptr_t)src % 16) || ((uintptr_t)dst % 8)) return CMDA_ERR_SHAPE;
  CMDA_DISPATCH_DTYPE(dst_dtype, CMDA_LAUNCH((cast_clear_kernel<T>), dim3(grid_for(n, 4)), dim3(256), 0, stream, src, (T*)dst, (long)(n / 4)));
  CMDA_CHECK_LAUNCH();
}

// desc: DEVICE array of `cmda_permute_desc_t` (include/cmda_hip.h); blocks: DEVICE int32 [nblocks][2] = {tensor index, chunk index}
// with one entry per 1024 destination elements of every tensor
// dst[r][0..cp) = cast(src[r][0..c)) | 0: fp32 rows of c columns -> activation-dtype rows padded to cp columns (the 19-class logit
// gradient -> 32 columns: its two GEMMs -- classifier weight gradient and data gradient -- then run on 16-byte chunks)
template <typename TD>
__global__ void cast_pad_cols_kernel(const float* __restrict__ src, TD* __restrict__ dst, long rows, int c, int cp) {
  const long total = rows * cp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = total < (1L << 32) ? (long)((unsigned)i / (unsigned)cp) : i / cp;   // (32-bit division: see sample_scale_kernel)
    const int k = (int)(i - r * cp);
    stf(dst + i, k < c ? src[r * c + k] : 0.f);
  }
}

extern "C" int cmda_cast_pad_cols(const float* src, void* dst, int64_t rows, int c, int cp, int dst_dtype, void* stream) {
  if (rows <= 0) return CMDA_OK;
  if (c <= 0 || cp < c) return CMDA_ERR_SHAPE;
  CMDA_DISPATCH_DTYPE(dst_dtype, CMDA_LAUNCH((cast_pad_cols_kernel<T>), dim3(grid_for(rows * cp, 1)), dim3(256), 0, stream, src, (T*)dst,
                                              (long)rows, c, cp));
  CMDA_CHECK_LAUNCH();
}

// out[r][c] = bias[c] (fp32): seeds the accumulator of a split-K GEMM whose epilogue cannot add the bias (atomic accumulation)
__global__ void rows_fill_kernel(float* __restrict__ out, const float* __restrict__ bias, long rows, int C) {
  const long n4 = rows * (long)(C >> 2);
  const int c4 = C >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) ld4(bias + c, v);
    st4(out + i * 4, v);
  }
}

extern "C" int cmda_rows_fill(float* out, const float* bias, int64_t rows, int C, void* stream) {
  if (rows <= 0) return CMDA_OK;
  if (C <= 0 || (C & 3)) return CMDA_ERR_SHAPE;
  CMDA_LAUNCH(rows_fill_kernel, dim3(grid_for(rows * (C >> 2), 1)), dim3(256), 0, stream, out, bias, (long)rows, C);
  CMDA_CHECK_LAUNCH();
}

extern "C" int cmda_nchw_to_nhwc_pad(const float* src, void* dst, int B, int C, int64_t HW, int cpad, int dst_dtype, void* stream) {
  if (B <= 0 || HW <= 0) return CMDA_OK;
  if (C <= 0 || cpad < C) return CMDA_ERR_SHAPE;
  const long n = (long)B * HW;
  CMDA_DISPATCH_DTYPE(dst_dtype, CMDA_LAUNCH((nchw_to_nhwc_pad_kernel<T>), dim3(grid_for(n, 1)), dim3(256), 0, stream, src, (T*)dst, B, C,
                                              (long)HW, cpad));
  CMDA_CHECK_LAUNCH();
}

extern "C" int cmda_permute4_batch(const void* desc, const int* blocks, int nblocks, void* stream) {
  if (nblocks <= 0) return CMDA_OK;
  static_assert(sizeof(PermuteDesc) == 64, "descriptor layout is part of the ABI");
  CMDA_LAUNCH(permute4_batch_kernel, dim3(nblocks), dim3(256), 0, stream, (const PermuteDesc*)desc, blocks);
  CMDA_CHECK_LAUNCH();
}

extern "C" int cmda_colsum(const void* x, float* out, int64_t M, int N, int64_t ld, int dtype, void* stream) {
  if (M <= 0 || N <= 0) return CMDA_OK;
  const int vec = ((ld & 3) == 0) && (((uintptr_t)x) % 16 == 0);
  int rpb = 512;
  const int gx = (N + 127) / 128;
  while (rpb > 64 && (M + rpb - 1) / rpb * gx < 256) rpb >>= 1;
  dim3 grid(gx, (unsigned)((M + rpb - 1) / rpb));
  CMDA_DISPATCH_DTYPE(dtype, CMDA_LAUNCH((colsum_kernel<T>), grid, dim3(256), 0, stream, (const T*)x, out, (long)M, N, (long)ld, rpb, vec));
  CMDA_CHECK_LAUNCH();
}

extern "C" int cmda_axpby(const void* x, const void* y, void* out, float a, float b, int64_t n, int dtype, void* stream) {
  if (n <= 0) return CMDA_OK;
  CMDA_DISPATCH_DTYPE(dtype, CMDA_LAUNCH((axpby_kernel<T>), dim3(grid_for(n, 4)), dim3(256), 0, stream, (const T*)x,
                                         (const T*)y, (T*)out, a, b, (long)n));
  CMDA_CHECK_LAUNCH();
}

// x/out: [B, per_sample] with per_sample = HW*C, C % 4 == 0; scale: [B] (per_channel=0) or [B,C] (per_channel=1)
extern "C" int cmda_sample_scale(const void* x, const float* scale, void* out, int B, int64_t per_sample, int C,
                                 int per_channel, int dtype, void* stream) {
  const long n = (long)B * per_sample;
  if (n <= 0) return CMDA_OK;
  if ((C & 3) || (per_sample % C)) return CMDA_ERR_SHAPE;
  CMDA_DISPATCH_DTYPE(dtype, CMDA_LAUNCH((sample_scale_kernel<T>), dim3(grid_for(n, 4)), dim3(256), 0, stream,
                                         (const T*)x, scale, (T*)out, (long)per_sample, C, per_channel, n));
  CMDA_CHECK_LAUNCH();
}

// src/dst already point at the first element; cols and both pitches must be multiples of 4
extern "C" int cmda_copy2d(const void* src, void* dst, int64_t rows, int cols, int64_t src_ld, int64_t dst_ld, int dtype,
                           void* stream) {
  if (rows <= 0 || cols <= 0) return CMDA_OK;
  if ((cols & 3) || (src_ld & 3) || (dst_ld & 3)) return CMDA_ERR_SHAPE;
  CMDA_DISPATCH_DTYPE(dtype, CMDA_LAUNCH((copy2d_kernel<T>), dim3(grid_for(rows * (cols / 4))), dim3(256), 0, stream,
                                         (const T*)src, (T*)dst, (long)rows, cols, (long)src_ld, (long)dst_ld));
  CMDA_CHECK_LAUNCH();
}

// Grid cap of the parameter-state passes (EMA, AdamW).  They stream at the HBM rate with far fewer workgroups than wave slots; what
// the cap decides is how much of the chip is left to kernels of OTHER streams meanwhile: with 4096 grid-stride workgroups every wave slot
// of every CU was held for the whole pass, and the mixing kernels / the generator's first convolutions that the overlapped step
// boundary runs beside them (optim.FlatAdamW.overlap) took 4-40 x their time (kernel trace, round 6: class_mix 531 us against 14).
static long stream_pass_blocks() {
  static const long v = [] {
    const char* e = getenv("CMDA_STREAM_BLOCKS");
    const long b = e ? atol(e) : 1024;
    return b > 0 ? b : 1024;
  }();
  return v;
}

extern "C" int cmda_ema_update(float* ema, const float* param, float alpha, int64_t n, void* ema_bf16, void* stream) {
  if (n <= 0) return CMDA_OK;
  // (non-temporal accesses: 243 -> 195 us over 85 M parameters, AdamW 472 -> 454; tools/hbm_bench.py, gpurun r04 A/B)
  static const bool nt = getenv("CMDA_STREAM_TEMPORAL") == nullptr;
  if (nt) CMDA_LAUNCH(ema_kernel<true>, dim3((unsigned)std::max<long>(1, std::min<long>((n + 2047) / 2048, stream_pass_blocks()))), dim3(256), 0, stream, ema, param, alpha,
              (long)n, (bf16_t*)ema_bf16);
  else CMDA_LAUNCH(ema_kernel<false>, dim3((unsigned)std::max<long>(1, std::min<long>((n + 2047) / 2048, stream_pass_blocks()))), dim3(256), 0, stream, ema, param, alpha,
              (long)n, (bf16_t*)ema_bf16);
  CMDA_CHECK_LAUNCH();
}

extern "C" int cmda_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr,
                               float beta1, float beta2, float eps, float weight_decay, int step, void* stream) {
  if (n <= 0) return CMDA_OK;
  if (step < 1) return CMDA_ERR_SHAPE;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = 1.f - powf(beta2, (float)step);
  const bool aligned = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0 && ((uintptr_t)p_bf16 & 7) == 0;
  const long n4 = aligned ? n / 4 : 0;
  static const bool nt = getenv("CMDA_STREAM_TEMPORAL") == nullptr;   // (A/B switch: plain loads / stores)
  if (n4 > 0 && nt)
    CMDA_LAUNCH(adamw_vec_kernel<true>, dim3((unsigned)std::max<long>(1, std::min<long>((n4 + 511) / 512, stream_pass_blocks()))), dim3(256), 0, stream, p, g, m,
                v, (bf16_t*)p_bf16, n4, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2));
  else if (n4 > 0)
    CMDA_LAUNCH(adamw_vec_kernel<false>, dim3((unsigned)std::max<long>(1, std::min<long>((n4 + 511) / 512, stream_pass_blocks()))), dim3(256), 0, stream, p, g, m,
                v, (bf16_t*)p_bf16, n4, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2));
  if (const long rest = n - 4 * n4) {
    bf16_t* pb = p_bf16 ? (bf16_t*)p_bf16 + 4 * n4 : nullptr;
    CMDA_LAUNCH(adamw_kernel, dim3(grid_for(rest)), dim3(256), 0, stream, p + 4 * n4, g + 4 * n4, m + 4 * n4, v + 4 * n4, pb, rest, lr,
                beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2));
  }
  CMDA_CHECK_LAUNCH();
}

// src/tgt/out: [B,Cch,H,W] (channels_last=0) or [B,H,W,Cch] (channels_last=1); src_label [B,H*W] int64;
// classes [B,max_classes] int64 padded with -1.
extern "C" int cmda_class_mix(const void* src, const void* tgt, void* out, const int64_t* src_label,
                              const int64_t* classes, int max_classes, int B, int HW, int Cch, int channels_last,
                              int dtype, void* stream) {
  const long total = (long)B * HW * Cch;
  if (total <= 0) return CMDA_OK;
  CMDA_DISPATCH_DTYPE(dtype, CMDA_LAUNCH((class_mix_kernel<T>), dim3(grid_for(total)), dim3(256), 0, stream,
                                         (const T*)src, (const T*)tgt, (T*)out, (const long long*)src_label,
                                         (const long long*)classes, max_classes, B, HW, Cch, channels_last));
  CMDA_CHECK_LAUNCH();
}

extern "C" int cmda_class_mix_label(const int64_t* src, const int64_t* tgt, int64_t* out, const int64_t* src_label,
                                    const int64_t* classes, int max_classes, int B, int HW, void* stream) {
  const long total = (long)B * HW;
  if (total <= 0) return CMDA_OK;
  CMDA_LAUNCH(class_mix_label_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (const long long*)src,
              (const long long*)tgt, (long long*)out, (const long long*)src_label, (const long long*)classes,
              max_classes, B, HW);
  CMDA_CHECK_LAUNCH();
}
