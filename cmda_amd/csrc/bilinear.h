// bilinear.h -- source index / weights of F.interpolate(mode='bilinear', align_corners=False) exactly as ATen's
// CPU path computes them in fp32 (area_pixel_compute_source_index + guard_index_and_lambda), shared by the resize,
// fused cross-entropy and pseudo-label kernels so that all of them sample identically.
// Reference: mmseg/ops/wrappers.py:9-28 (resize -> F.interpolate); align_corners is False everywhere on the path
// (decode_heads/daformer_head.py:207 asserts it).
#pragma once
#include "common.h"

struct BilinTap {
  int i0, i1;
  float l0, l1;
};

static __device__ __forceinline__ BilinTap bilin_tap(int dst, int in_size, int out_size, float scale) {
#pragma clang fp contract(off)
  BilinTap t;
  if (in_size == out_size) {
    t.i0 = t.i1 = dst;
    t.l0 = 1.f;
    t.l1 = 0.f;
    return t;
  }
  float real = scale * ((float)dst + 0.5f) - 0.5f;
  if (real < 0.f) real = 0.f;
  int i0 = (int)floorf(real);
  if (i0 > in_size - 1) i0 = in_size - 1;
  float lam = real - (float)i0;
  lam = fminf(fmaxf(lam, 0.f), 1.f);
  t.i0 = i0;
  t.i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  t.l1 = lam;
  t.l0 = 1.f - lam;
  return t;
}

// value = (v00*wx0 + v01*wx1)*wy0 + (v10*wx0 + v11*wx1)*wy1, every product and sum individually rounded
static __device__ __forceinline__ float bilin_mix(float v00, float v01, float v10, float v11, float wx0, float wx1,
                                                  float wy0, float wy1) {
#pragma clang fp contract(off)
  const float a = v00 * wx0;
  const float b = v01 * wx1;
  const float top = a + b;
  const float c = v10 * wx0;
  const float d = v11 * wx1;
  const float bot = c + d;
  const float e = top * wy0;
  const float f = bot * wy1;
  return e + f;
}
