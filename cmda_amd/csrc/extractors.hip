// extractors.hip -- the two "extractor" front-ends of CMDA that the reference runs on the CPU inside the step/loader.
//
// (1) Image Content-Extractor (ISR): mmseg/datasets/utils.py get_ic :87-105 + get_image_change_from_pil :108-152, called
//     in the training step on the ClassMix-ed image (uda/dacs.py:729-744: GPU->CPU copy -> PIL 'L' -> numpy log-diff ->
//     torch min/max -> back to the GPU, one host round trip + sync per sample).  Here it stays on the device:
//       isr_gray   : denorm, clamp, *255, uint8 truncation, PIL's integer luma (19595 R + 38470 G + 7471 B + 0x8000) >> 16
//       isr_minmax : log-intensity difference against the shifted image (256-entry LUT of the reference's own
//                    float32 log values), dead-zone, clip, global min/max of the + / - parts (per sample, per direction)
//       isr_apply  : separate min-max normalisation of the two parts, average of the row / column variants, x3 channels
// (2) Event voxel grid: mmseg/datasets/dsec.py events_to_voxel_grid :26-70 (tri-linear scatter-add of polarity) and
//     events_norm :80-121 (non-zero standardise, clip, +/- min-max).
// All HBM/atomic-bound; fp32 throughout; min/max via integer atomics on the (non-negative) float bit patterns.
#include "common.h"

namespace {

// min/max scratch: per record {min=+inf, max=0, min=+inf, max=0} as float bit patterns (non-negative floats order like uints)
__global__ void minmax_init_kernel(unsigned* __restrict__ mm, int nrec) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nrec * 4) mm[i] = (i & 1) ? 0u : 0x7F800000u;
}

// ---------------------------------------------------------------------------------------------------- ISR
// img: NCHW fp32 [B,3,H,W] (normalised), gray: uint8 [B,H,W]
__global__ void isr_gray_kernel(const float* __restrict__ img, unsigned char* __restrict__ gray, int B, int HW,
                                float m0, float m1, float m2, float s0, float s1, float s2) {
#pragma clang fp contract(off)
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW, p = i - b * HW;
    const float* px = img + b * 3 * HW + p;
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
    unsigned u[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = px[(long)c * HW] * stdv[c];
      v = v + mean[c];
      v = v / 255.0f;
      v = fminf(fmaxf(v, 0.f), 1.f);
      v = v * 255.f;
      u[c] = (unsigned)v;  // np.uint8(): truncation
    }
    gray[i] = (unsigned char)((19595u * u[0] + 38470u * u[1] + 7471u * u[2] + 0x8000u) >> 16);
  }
}

static __device__ __forceinline__ float isr_diff(const unsigned char* __restrict__ g, const float* __restrict__ lut,
                                                 int y, int x, int H, int W, int dy, int dx, float thr) {
  // shifted copy with "edge = itself" semantics of np.concatenate in get_image_change_from_pil
  const int sy = y + dy, sx = x + dx;
  const bool inside = sy >= 0 && sy < H && sx >= 0 && sx < W;
  const float front = lut[g[y * W + x]];
  const float now = inside ? lut[g[sy * W + sx]] : front;
  const float d = now - front;
  return fabsf(d) <= thr ? 0.f : d;
}

// mm[b][dir][4] (as uint bit patterns of non-negative floats): pos_min, pos_max, negabs_min, negabs_max
__global__ void isr_minmax_kernel(const unsigned char* __restrict__ gray, const float* __restrict__ lut,
                                  unsigned* __restrict__ mm, int H, int W, int ndir, const int* __restrict__ dirs,
                                  float thr, float clip) {
  __shared__ unsigned red[4][4];
  const int b = blockIdx.y;
  const int dir = blockIdx.z;
  const int dy = dirs[dir * 2], dx = dirs[dir * 2 + 1];
  const unsigned char* g = gray + (long)b * H * W;
  float pmin = INFINITY, pmax = 0.f, nmin = INFINITY, nmax = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * W; i += gridDim.x * blockDim.x) {
    const int y = i / W, x = i - y * W;
    const float d = isr_diff(g, lut, y, x, H, W, dy, dx, thr);
    const float pos = fminf(fmaxf(d, 0.f), clip);
    const float na = fminf(fmaxf(-d, 0.f), clip);  // |negative part|
    pmin = fminf(pmin, pos); pmax = fmaxf(pmax, pos);
    nmin = fminf(nmin, na); nmax = fmaxf(nmax, na);
  }
  pmin = wave_min(pmin); pmax = wave_max(pmax); nmin = wave_min(nmin); nmax = wave_max(nmax);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) {
    red[wid][0] = __float_as_uint(pmin); red[wid][1] = __float_as_uint(pmax);
    red[wid][2] = __float_as_uint(nmin); red[wid][3] = __float_as_uint(nmax);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned a = red[0][0], bq = red[0][1], c = red[0][2], dd = red[0][3];
    for (int w = 1; w < 4; ++w) {
      a = min(a, red[w][0]); bq = max(bq, red[w][1]); c = min(c, red[w][2]); dd = max(dd, red[w][3]);
    }
    unsigned* o = mm + ((long)b * ndir + dir) * 4;
    atomicMin(o + 0, a); atomicMax(o + 1, bq); atomicMin(o + 2, c); atomicMax(o + 3, dd);
  }
}

// out: NCHW fp32 [B,3,H,W]
__global__ void isr_apply_kernel(const unsigned char* __restrict__ gray, const float* __restrict__ lut,
                                 const unsigned* __restrict__ mm, float* __restrict__ out, int B, int H, int W, int ndir,
                                 const int* __restrict__ dirs, float thr, float clip) {
#pragma clang fp contract(off)
  const long total = (long)B * H * W;
  const float share = 1.f / (float)ndir;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / ((long)H * W));
    const int p = (int)(i - (long)b * H * W);
    const int y = p / W, x = p - y * W;
    const unsigned char* g = gray + (long)b * H * W;
    float acc = 0.f;
    for (int dir = 0; dir < ndir; ++dir) {
      const unsigned* m = mm + ((long)b * ndir + dir) * 4;
      const float pmin = __uint_as_float(m[0]), pmax = __uint_as_float(m[1]);
      const float namin = __uint_as_float(m[2]), namax = __uint_as_float(m[3]);
      const float d = isr_diff(g, lut, y, x, H, W, dirs[dir * 2], dirs[dir * 2 + 1], thr);
      const float pos = fminf(fmaxf(d, 0.f), clip);
      const float neg = fminf(fmaxf(d, -clip), 0.f);
      // tensor_normalize_to_range: (t - tmin) / (tmax - tmin + 1e-8) * (hi - lo) + lo
      const float pn = (pos - pmin) / (pmax - pmin + 1e-8f) * 1.f + 0.f;
      const float nlo = -namax, nhi = -namin;  // min / max of the negative part
      const float nn = (neg - nlo) / (nhi - nlo + 1e-8f) * 1.f + -1.f;
      acc += (pn + nn) * share;
    }
    float* o = out + (long)b * 3 * H * W + p;
    o[0] = acc; o[(long)H * W] = acc; o[2L * H * W] = acc;
  }
}

// ---------------------------------------------------------------------------------------------------- voxel grid
__global__ void voxel_scatter_kernel(const float* __restrict__ t, const float* __restrict__ x, const float* __restrict__ y,
                                     const float* __restrict__ pol, float* __restrict__ grid, long N, int C, int H, int W) {
#pragma clang fp contract(off)
  const float t0 = t[0], tN = t[N - 1];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
    const float tn = (float)(C - 1) * (t[i] - t0) / (tN - t0);
    const float xv = x[i], yv = y[i];
    const int x0 = (int)xv, y0 = (int)yv, tb = (int)tn;
    const float value = 2.f * pol[i] - 1.f;
#pragma unroll
    for (int ax = 0; ax < 2; ++ax)
#pragma unroll
      for (int ay = 0; ay < 2; ++ay)
#pragma unroll
        for (int at = 0; at < 2; ++at) {
          const int xl = x0 + ax, yl = y0 + ay, tl = tb + at;
          if (xl < W && xl >= 0 && yl < H && yl >= 0 && tl >= 0 && tl < C) {
            const float wgt = value * (1.f - fabsf((float)xl - xv)) * (1.f - fabsf((float)yl - yv)) *
                              (1.f - fabsf((float)tl - tn));
            atomicAdd(grid + ((long)tl * H + yl) * W + xl, wgt);
          }
        }
  }
}

// ws (double[3]): count of non-zeros, sum, sum of squares
__global__ void events_stats_kernel(const float* __restrict__ e, double* __restrict__ ws, long n) {
  __shared__ double buf[256][3];
  double c = 0, s = 0, q = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = e[i];
    if (v != 0.f) c += 1.0;
    s += v;
    q += (double)v * v;
  }
  buf[threadIdx.x][0] = c; buf[threadIdx.x][1] = s; buf[threadIdx.x][2] = q;
  __syncthreads();
  if (threadIdx.x < 3) {
    double a = 0;
    for (int k = 0; k < (int)blockDim.x; ++k) a += buf[k][threadIdx.x];
    atomicAdd(ws + threadIdx.x, a);
  }
}

static __device__ __forceinline__ float events_standardise(float v, float mean, float stdv, int any) {
  if (!any) return v;
  return (v != 0.f ? 1.f : 0.f) * (v - mean) / (stdv + 1e-8f);
}

// mm (uint[4]): pos_min, pos_max, negabs_min, negabs_max
__global__ void events_minmax_kernel(const float* __restrict__ e, const double* __restrict__ ws, unsigned* __restrict__ mm,
                                     long n, float clip) {
  __shared__ unsigned red[4][4];
  const double cnt = ws[0];
  const int any = cnt > 0;
  const float mean = any ? (float)(ws[1] / cnt) : 0.f;
  const float stdv = any ? sqrtf((float)(ws[2] / cnt) - mean * mean) : 1.f;
  float pmin = INFINITY, pmax = 0.f, nmin = INFINITY, nmax = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = events_standardise(e[i], mean, stdv, any);
    const float pos = fminf(fmaxf(v, 0.f), clip), na = fminf(fmaxf(-v, 0.f), clip);
    pmin = fminf(pmin, pos); pmax = fmaxf(pmax, pos); nmin = fminf(nmin, na); nmax = fmaxf(nmax, na);
  }
  pmin = wave_min(pmin); pmax = wave_max(pmax); nmin = wave_min(nmin); nmax = wave_max(nmax);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) {
    red[wid][0] = __float_as_uint(pmin); red[wid][1] = __float_as_uint(pmax);
    red[wid][2] = __float_as_uint(nmin); red[wid][3] = __float_as_uint(nmax);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned a = red[0][0], b = red[0][1], c = red[0][2], d = red[0][3];
    for (int w = 1; w < 4; ++w) { a = min(a, red[w][0]); b = max(b, red[w][1]); c = min(c, red[w][2]); d = max(d, red[w][3]); }
    atomicMin(mm + 0, a); atomicMax(mm + 1, b); atomicMin(mm + 2, c); atomicMax(mm + 3, d);
  }
}

__global__ void events_apply_kernel(const float* __restrict__ e, const double* __restrict__ ws,
                                    const unsigned* __restrict__ mm, float* __restrict__ out, long n, float clip,
                                    float final_range) {
#pragma clang fp contract(off)
  const double cnt = ws[0];
  const int any = cnt > 0;
  const float mean = any ? (float)(ws[1] / cnt) : 0.f;
  const float stdv = any ? sqrtf((float)(ws[2] / cnt) - mean * mean) : 1.f;
  const float pmin = __uint_as_float(mm[0]), pmax = __uint_as_float(mm[1]);
  const float nlo = -__uint_as_float(mm[3]), nhi = -__uint_as_float(mm[2]);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = events_standardise(e[i], mean, stdv, any);
    const float pos = fminf(fmaxf(v, 0.f), clip), neg = fminf(fmaxf(v, -clip), 0.f);
    const float pn = (pos - pmin) / (pmax - pmin + 1e-8f) * (final_range - 0.f) + 0.f;
    const float nn = (neg - nlo) / (nhi - nlo + 1e-8f) * (0.f - -final_range) + -final_range;
    out[i] = pn + nn;
  }
}

static inline int grid_for(long n) { return (int)std::max<long>(1, std::min<long>((n + 255) / 256, 4096)); }
}  // namespace

// img NCHW fp32 [B,3,H,W]; gray uint8 [B,H,W]; mean3/std3: HOST pointers to the 3 normalisation constants
extern "C" int cmda_isr_gray(const float* img, uint8_t* gray, int B, int H, int W, const float* mean3, const float* std3,
                             void* stream) {
  if ((long)B * H * W <= 0) return CMDA_OK;
  CMDA_LAUNCH(isr_gray_kernel, dim3(grid_for((long)B * H * W)), dim3(256), 0, stream, img, (unsigned char*)gray, B, H * W,
              mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  CMDA_CHECK_LAUNCH();
}

// lut: device fp32[256] = log(L/255*(v1-v0)+v0); dirs: device int32[ndir*2] (dy,dx of the shifted copy);
// mm: device uint32[B*ndir*4] scratch (initialised here); out NCHW fp32 [B,3,H,W].
extern "C" int cmda_isr_from_gray(const uint8_t* gray, const float* lut, const int* dirs, int ndir, uint32_t* mm, float* out,
                                  int B, int H, int W, float threshold, float clip, void* stream) {
  if ((long)B * H * W <= 0) return CMDA_OK;
  if (ndir <= 0 || ndir > 4) return CMDA_ERR_SHAPE;
  CMDA_LAUNCH(minmax_init_kernel, dim3((B * ndir * 4 + 255) / 256), dim3(256), 0, stream, (unsigned*)mm, B * ndir);
  dim3 grid(std::max(1, std::min((H * W + 255) / 256, 256)), B, ndir);
  CMDA_LAUNCH(isr_minmax_kernel, grid, dim3(256), 0, stream, (const unsigned char*)gray, lut, (unsigned*)mm, H, W, ndir,
              dirs, threshold, clip);
  CMDA_LAUNCH(isr_apply_kernel, dim3(grid_for((long)B * H * W)), dim3(256), 0, stream, (const unsigned char*)gray, lut,
              (const unsigned*)mm, out, B, H, W, ndir, dirs, threshold, clip);
  CMDA_CHECK_LAUNCH();
}

// events as 4 fp32 arrays of length N (t sorted); grid fp32 [bins,H,W] (zeroed here)
extern "C" int cmda_events_to_voxel_grid(const float* t, const float* x, const float* y, const float* pol, float* grid,
                                         int64_t N, int bins, int H, int W, void* stream) {
  cmda_zero_async(grid, sizeof(float) * (size_t)bins * H * W, stream);
  if (N <= 0) return CMDA_OK;
  CMDA_LAUNCH(voxel_scatter_kernel, dim3(grid_for(N)), dim3(256), 0, stream, t, x, y, pol, grid, (long)N, bins, H, W);
  CMDA_CHECK_LAUNCH();
}

// ws: device scratch of 3 doubles + 4 uint32 (40 bytes, 8-byte aligned); out may alias events
extern "C" int cmda_events_norm(const float* events, float* out, void* ws, int64_t n, float clip_range, float final_range,
                                void* stream) {
  if (n <= 0) return CMDA_OK;
  double* dws = (double*)ws;
  uint32_t* mm = (uint32_t*)(dws + 3);
  cmda_zero_async(dws, 3 * sizeof(double), stream);
  CMDA_LAUNCH(minmax_init_kernel, dim3(1), dim3(256), 0, stream, (unsigned*)mm, 1);
  CMDA_LAUNCH(events_stats_kernel, dim3(grid_for(n)), dim3(256), 0, stream, events, dws, (long)n);
  CMDA_LAUNCH(events_minmax_kernel, dim3(grid_for(n)), dim3(256), 0, stream, events, (const double*)dws, (unsigned*)mm,
              (long)n, clip_range);
  CMDA_LAUNCH(events_apply_kernel, dim3(grid_for(n)), dim3(256), 0, stream, events, (const double*)dws,
              (const unsigned*)mm, out, (long)n, clip_range, final_range);
  CMDA_CHECK_LAUNCH();
}
