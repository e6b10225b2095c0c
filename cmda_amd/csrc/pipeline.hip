// pipeline.hip -- the loader-side preprocessing of CMDA's two datasets as device kernels: what
// mmseg/datasets/dsec.py:189-339 (__getitem__) and :341-366 (get_events_vg) do per TARGET sample on CPU workers with PIL / numpy /
// h5py, the SOURCE sample's resize -> crop -> flip of mmseg/datasets/cityscapes_ic.py:147-210, and the offline Motion-Extractor
// pre-step create_cityscapes_image_change.py:16-35 (log-intensity time residual of two consecutive frames).
//
//   pil_resample_{h,v}  : PIL.Image.resize(resample=BILINEAR) on uint8 frames, bit for bit -- Pillow's two-pass fixed-point
//                         convolution (third-party Pillow 8.3.1, src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc
//                         with 22 fraction bits, horizontal pass rounded to uint8, then the vertical pass; published algorithm,
//                         restated; the coefficient tables are built on the host in double precision exactly as Pillow builds
//                         them, cmda_amd/pipeline.py).  Crop boxes and horizontal flips of the loaders are address arithmetic:
//                         per-sample offsets / flags come from a DEVICE array, so one captured launch serves every sample.
//                         The vertical pass writes uint8 HWC and / or the normalised float NCHW tensor (ToTensor + Normalize, or
//                         (x/255 - 0.5)/0.5 repeated to 3 channels) and / or PIL's 'L' luma of the result (input of the ISR).
//   time_residual_*     : get_image_change(): dead-zone / clip / separate min-max of the + and - parts of log(now+c) - log(front+c),
//                         quantised to uint8 with round-half-even, as the PNG the source loader later reads.
//   event_prep          : rectification gather xy = rectify_map[y, x] and t -> (t - t0) / (tN - t0) in fp32 (dsec.py:341-353).
//   crop_flip_resize    : events_vg[:, y:y+ch, x:x+cw] -> flip -> F.interpolate(bilinear, align_corners=False) -> x3 (dsec.py:314-322).
// All HBM-bound byte / gather kernels: coalesced along the pixel's channel-interleaved row, no LDS needed at these sizes
// (a 480x640x3 frame is 0.9 MB, L2-resident between the two passes).
#include "bilinear.h"
#include "common.h"

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;   // Pillow: PRECISION_BITS

static __device__ __forceinline__ unsigned char clip8(int v) {
  v >>= kPrecisionBits;                       // arithmetic shift, as Pillow's lookup index
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// samp[b] = {src_x0, src_y0, flip_src, out_x0, out_y0, flip_out, 0, 0}
//   source pixel (x, y) of the image the resize sees = frame[src_y0 + y][src_x0 + (flip_src ? in_w-1-x : x)]
//   output pixel (ox, oy) = resized[out_y0 + oy][out_x0 + (flip_out ? OW-1-ox : ox)]
struct Samp { int src_x0, src_y0, flip_src, out_x0, out_y0, flip_out, pad0, pad1; };

// horizontal pass: tmp[b][y][ox][c], y over the in_h rows of the (cropped) input, ox over the OW output-window columns
__global__ void pil_resample_h_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ tmp,
                                      const Samp* __restrict__ samp, const int* __restrict__ bounds, const int* __restrict__ kk,
                                      int ksize, int B, int IH, int IW, int C, int in_w, int in_h, int OW) {
  const long total = (long)B * in_h * OW * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long t = i / C;
    const int ox = (int)(t % OW);
    t /= OW;
    const int y = (int)(t % in_h);
    const int b = (int)(t / in_h);
    const Samp s = samp[b];
    const int rx = s.out_x0 + (s.flip_out ? OW - 1 - ox : ox);
    const int xmin = bounds[rx * 2], xmax = bounds[rx * 2 + 1];
    const int* k = kk + (long)rx * ksize;
    const unsigned char* row = src + ((long)b * IH + (s.src_y0 + y)) * IW * C;
    int ss = 1 << (kPrecisionBits - 1);
    for (int x = 0; x < xmax; ++x) {
      const int xs = x + xmin;
      const int sx = s.src_x0 + (s.flip_src ? in_w - 1 - xs : xs);
      ss += (int)row[(long)sx * C + c] * k[x];
    }
    tmp[i] = clip8(ss);
  }
}

// vertical pass over tmp [B][in_h][OW][C] -> out window rows; optional outputs (any may be null):
//   out_u8 [B][OH][OW][C];  out_f NCHW [B][Cout][OH][OW] = u8 * fscale[c] + fshift[c] (Cout = 3 when C == 1 and rep3);
//   out_gray [B][OH][OW] = PIL 'L' luma of the RGB result
__global__ void pil_resample_v_kernel(const unsigned char* __restrict__ tmp, const Samp* __restrict__ samp,
                                      const int* __restrict__ bounds, const int* __restrict__ kk, int ksize, int B, int C, int in_h,
                                      int OW, int OH, unsigned char* __restrict__ out_u8, float* __restrict__ out_f,
                                      unsigned char* __restrict__ out_gray, float a0, float a1, float a2, float b0, float b1,
                                      float b2, int rep3) {
#pragma clang fp contract(off)
  const long total = (long)B * OH * OW;
  const float fa[3] = {a0, a1, a2}, fb[3] = {b0, b1, b2};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    long t = i / OW;
    const int oy = (int)(t % OH);
    const int b = (int)(t / OH);
    const Samp s = samp[b];
    const int ry = s.out_y0 + oy;
    const int ymin = bounds[ry * 2], ymax = bounds[ry * 2 + 1];
    const int* k = kk + (long)ry * ksize;
    unsigned char px[4] = {0, 0, 0, 0};
    for (int c = 0; c < C; ++c) {
      int ss = 1 << (kPrecisionBits - 1);
      for (int y = 0; y < ymax; ++y) ss += (int)tmp[(((long)b * in_h + (y + ymin)) * OW + ox) * C + c] * k[y];
      px[c] = clip8(ss);
    }
    if (out_u8)
      for (int c = 0; c < C; ++c) out_u8[i * C + c] = px[c];
    if (out_f) {
      const int Cout = (C == 1 && rep3) ? 3 : C;
      for (int c = 0; c < Cout; ++c) {
        const float v = (float)px[C == 1 ? 0 : c];
        // ToTensor: u8 / 255 ; Normalize: (t - mean) / std -- kept as the two rounded steps of the reference
        const float tt = v / 255.0f;
        out_f[(((long)b * Cout + c) * OH + oy) * OW + ox] = (tt - fb[c]) / fa[c];
      }
    }
    if (out_gray) out_gray[i] = (unsigned char)((19595u * px[0] + 38470u * px[1] + 7471u * px[2] + 0x8000u) >> 16);
  }
}

// ---------------------------------------------------------------------------------------------------- time residual
__global__ void pair_minmax_init_kernel(unsigned* __restrict__ mm, int nrec) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nrec * 4) mm[i] = (i & 1) ? 0u : 0x7F800000u;
}

static __device__ __forceinline__ float pair_diff(const unsigned char* __restrict__ now, const unsigned char* __restrict__ front,
                                                  const float* __restrict__ lut, long i, float thr) {
  const float d = lut[now[i]] - lut[front[i]];
  return fabsf(d) <= thr ? 0.f : d;
}

__global__ void pair_minmax_kernel(const unsigned char* __restrict__ now, const unsigned char* __restrict__ front,
                                   const float* __restrict__ lut, unsigned* __restrict__ mm, long HW, float thr, float clip) {
  __shared__ unsigned red[4][4];
  const int b = blockIdx.y;
  float pmin = INFINITY, pmax = 0.f, nmin = INFINITY, nmax = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x) {
    const float d = pair_diff(now + b * HW, front + b * HW, lut, i, thr);
    const float pos = fminf(fmaxf(d, 0.f), clip), na = fminf(fmaxf(-d, 0.f), clip);
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
    unsigned* o = mm + (long)b * 4;
    atomicMin(o + 0, a); atomicMax(o + 1, bq); atomicMin(o + 2, c); atomicMax(o + 3, dd);
  }
}

// out u8 = uint8(around((v + 1) / 2 * 255)), v = pos-part normalised to [0,1] + neg-part normalised to [-1,0]
__global__ void pair_apply_kernel(const unsigned char* __restrict__ now, const unsigned char* __restrict__ front,
                                  const float* __restrict__ lut, const unsigned* __restrict__ mm, unsigned char* __restrict__ out,
                                  int B, long HW, float thr, float clip) {
#pragma clang fp contract(off)
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW);
    const unsigned* m = mm + (long)b * 4;
    const float pmin = __uint_as_float(m[0]), pmax = __uint_as_float(m[1]);
    const float namin = __uint_as_float(m[2]), namax = __uint_as_float(m[3]);
    const float d = pair_diff(now, front, lut, i, thr);
    const float pos = fminf(fmaxf(d, 0.f), clip);
    const float neg = fminf(fmaxf(d, -clip), 0.f);
    const float pn = (pos - pmin) / (pmax - pmin + 1e-8f) * 1.f + 0.f;
    const float nlo = -namax, nhi = -namin;
    const float nn = (neg - nlo) / (nhi - nlo + 1e-8f) * 1.f + -1.f;
    float v = pn + nn;
    v = v + 1.f;
    v = v / 2.f;
    v = v * 255.f;
    out[i] = (unsigned char)rintf(v);   // np.around: round half to even
  }
}

// ---------------------------------------------------------------------------------------------------- events
__global__ void event_prep_kernel(const long long* __restrict__ t, const int* __restrict__ x, const int* __restrict__ y,
                                  const unsigned char* __restrict__ p, const float* __restrict__ rect_map, int H, int W,
                                  float* __restrict__ tn, float* __restrict__ xr, float* __restrict__ yr, float* __restrict__ pol,
                                  long N) {
#pragma clang fp contract(off)
  const long long t0 = t[0];
  const float tlast = (float)(t[N - 1] - t0);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
    tn[i] = (float)(t[i] - t0) / tlast;
    pol[i] = (float)p[i];
    const int xi = x[i], yi = y[i];
    if (rect_map) {
      const float* m = rect_map + ((long)yi * W + xi) * 2;
      xr[i] = m[0];
      yr[i] = m[1];
    } else {
      xr[i] = (float)xi;
      yr[i] = (float)yi;
    }
  }
}

// in [B][C][IH][IW] fp32 -> out [B][C*rep][OH][OW]: crop window (x0,y0,cw,ch) per sample, optional horizontal flip, bilinear
__global__ void crop_flip_resize_kernel(const float* __restrict__ in, float* __restrict__ out, const Samp* __restrict__ samp,
                                        int B, int C, int IH, int IW, int cw, int ch, int OH, int OW, int rep) {
  const long total = (long)B * C * OH * OW;
  const float sy = (float)ch / (float)OH, sx = (float)cw / (float)OW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    long t = i / OW;
    const int oy = (int)(t % OH);
    t /= OH;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    const Samp s = samp[b];
    const BilinTap ty = bilin_tap(oy, ch, OH, sy), tx = bilin_tap(ox, cw, OW, sx);
    const float* pl = in + ((long)b * C + c) * IH * IW;
    auto at = [&](int yy, int xx) {
      const int xs = s.flip_src ? cw - 1 - xx : xx;
      return pl[(long)(s.src_y0 + yy) * IW + s.src_x0 + xs];
    };
    const float v = bilin_mix(at(ty.i0, tx.i0), at(ty.i0, tx.i1), at(ty.i1, tx.i0), at(ty.i1, tx.i1), tx.l0, tx.l1, ty.l0, ty.l1);
    for (int r = 0; r < rep; ++r) out[(((long)b * C * rep + (long)r * C + c) * OH + oy) * OW + ox] = v;
  }
}

// PIL Image.convert('L') of uint8 RGB pixels (interleaved): L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16
__global__ void luma_u8_kernel(const unsigned char* __restrict__ rgb, unsigned char* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const unsigned char* p = rgb + i * 3;
    out[i] = (unsigned char)((19595u * p[0] + 38470u * p[1] + 7471u * p[2] + 0x8000u) >> 16);
  }
}

static inline int grid_for(long n) { return (int)std::max<long>(1, std::min<long>((n + 255) / 256, 16384)); }
}  // namespace

extern "C" int cmda_luma_u8(const uint8_t* rgb, uint8_t* out, int64_t npix, void* stream) {
  if (npix <= 0) return CMDA_OK;
  CMDA_LAUNCH(luma_u8_kernel, dim3(grid_for(npix)), dim3(256), 0, stream, rgb, out, (long)npix);
  CMDA_CHECK_LAUNCH();
}

extern "C" int cmda_pil_resize_u8(const uint8_t* src, int B, int IH, int IW, int C, const int* samp, int in_w, int in_h,
                                  const int* hbounds, const int* hkk, int hksize, const int* vbounds, const int* vkk, int vksize,
                                  int OW, int OH, uint8_t* tmp, uint8_t* out_u8, float* out_f, uint8_t* out_gray,
                                  const float* fscale3, const float* fshift3, int rep3, void* stream) {
  if (B <= 0 || OW <= 0 || OH <= 0) return CMDA_OK;
  if (C < 1 || C > 4 || in_w > IW || in_h > IH || (out_gray && C != 3) || (out_f && C != 1 && C != 3)) return CMDA_ERR_SHAPE;
  CMDA_LAUNCH(pil_resample_h_kernel, dim3(grid_for((long)B * in_h * OW * C)), dim3(256), 0, stream, src, tmp, (const Samp*)samp,
              hbounds, hkk, hksize, B, IH, IW, C, in_w, in_h, OW);
  const float one[3] = {1.f, 1.f, 1.f}, zero[3] = {0.f, 0.f, 0.f};
  const float* a = fscale3 ? fscale3 : one;
  const float* b = fshift3 ? fshift3 : zero;
  CMDA_LAUNCH(pil_resample_v_kernel, dim3(grid_for((long)B * OH * OW)), dim3(256), 0, stream, (const unsigned char*)tmp,
              (const Samp*)samp, vbounds, vkk, vksize, B, C, in_h, OW, OH, out_u8, out_f, out_gray, a[0], a[1], a[2], b[0], b[1],
              b[2], rep3);
  CMDA_CHECK_LAUNCH();
}

extern "C" int cmda_time_residual_u8(const uint8_t* now, const uint8_t* front, const float* lut, uint32_t* mm, uint8_t* out, int B,
                                     int H, int W, float threshold, float clip, void* stream) {
  const long HW = (long)H * W;
  if (B <= 0 || HW <= 0) return CMDA_OK;
  CMDA_LAUNCH(pair_minmax_init_kernel, dim3((B * 4 + 255) / 256), dim3(256), 0, stream, mm, B);
  CMDA_LAUNCH(pair_minmax_kernel, dim3((unsigned)std::min<long>((HW + 255) / 256, 512), B), dim3(256), 0, stream, now, front, lut,
              mm, HW, threshold, clip);
  CMDA_LAUNCH(pair_apply_kernel, dim3(grid_for((long)B * HW)), dim3(256), 0, stream, now, front, lut, (const unsigned*)mm, out, B,
              HW, threshold, clip);
  CMDA_CHECK_LAUNCH();
}

extern "C" int cmda_event_prep(const int64_t* t, const int32_t* x, const int32_t* y, const uint8_t* p, const float* rect_map, int H,
                               int W, float* t_norm, float* xr, float* yr, float* pol, int64_t N, void* stream) {
  if (N <= 0) return CMDA_OK;
  CMDA_LAUNCH(event_prep_kernel, dim3(grid_for(N)), dim3(256), 0, stream, (const long long*)t, x, y, p, rect_map, H, W, t_norm, xr,
              yr, pol, (long)N);
  CMDA_CHECK_LAUNCH();
}

extern "C" int cmda_crop_flip_resize_f32(const float* in, float* out, const int* samp, int B, int C, int IH, int IW, int cw, int ch,
                                         int OH, int OW, int rep, void* stream) {
  if (B <= 0 || C <= 0) return CMDA_OK;
  if (cw > IW || ch > IH || rep < 1) return CMDA_ERR_SHAPE;
  CMDA_LAUNCH(crop_flip_resize_kernel, dim3(grid_for((long)B * C * OH * OW)), dim3(256), 0, stream, in, out, (const Samp*)samp, B,
              C, IH, IW, cw, ch, OH, OW, rep);
  CMDA_CHECK_LAUNCH();
}
