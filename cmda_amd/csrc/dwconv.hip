// dwconv.hip -- depthwise 3x3 convolution (optionally dilated) on NHWC activations.
//
// Reference ops replaced:
//   DWConv (3x3, pad 1, bias, groups=C) + GELU inside MixFFN   mix_transformer.py:37-44,443-455
//   depthwise half of DepthwiseSeparableConvModule (3x3, dilation 6/12/18, no bias) in the sep-ASPP
//   decode_heads/sep_aspp_head.py:18-27 (mmcv DepthwiseSeparableConvModule.depthwise_conv)
//
// HBM-bound stencils: a thread owns 4 adjacent channels and a RUN of 8 pixels of one image row spaced `dil` apart; it
// slides a 3x3 register window along the run, so every output costs 3 new coalesced 8/16-byte loads (lanes run along
// C) instead of 9.  Algorithmic bytes per pixel-channel:
// fwd 2*sizeof(T); gelu-bwd-prep 3*sizeof(T); bwd-data 2*sizeof(T); bwd-weight 2*sizeof(T).
// Depthwise weights/bias stay fp32.  The stencil kernels read a tap-major [9][C] copy of the reference's [C,1,3,3]
// parameter (rt.wdw: one tiny permute per optimizer step) so that the 4 channel weights of a lane are one coalesced
// 16-byte load per tap; the weight-gradient kernel still accumulates in the parameter's own [C,9] layout.
#include "common.h"

namespace {

constexpr int kRun = 8;  // outputs per thread along a row (stride dil)

// Runs: a row's columns split into `dil` residue classes; class rho holds w = rho, rho+dil, ... and is cut into runs
// of kRun outputs.  run id -> (b, h, rho, k); first column w0 = rho + k*kRun*dil.
struct RunGeom {
  int H, W, dil, rpc, rpr;  // rpc: runs per residue class, rpr = dil * rpc runs per row
  long nruns;
};
static inline RunGeom run_geom(int B, int H, int W, int dil) {
  RunGeom g;
  g.H = H; g.W = W; g.dil = dil;
  const int per_class = (W + dil - 1) / dil;
  g.rpc = (per_class + kRun - 1) / kRun;
  g.rpr = dil * g.rpc;
  g.nruns = (long)B * H * g.rpr;
  return g;
}

// column of three taps (rows h-dil, h, h+dil) at column iw for 4 channels; zero outside the image
template <typename T>
static __device__ __forceinline__ void load_col(const T* __restrict__ x, long row_base, int h, int iw, int H, int W, int C,
                                                int dil, float col[3][4]) {
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int ih = h + (kh - 1) * dil;
    if (iw >= 0 && iw < W && ih >= 0 && ih < H) {
      ld4(x + (row_base + (long)(kh - 1) * dil * W + iw) * C, col[kh]);
    } else {
      col[kh][0] = col[kh][1] = col[kh][2] = col[kh][3] = 0.f;
    }
  }
}

// Stencil kernels: block = 64 channel-quads x 4 run lanes; a thread keeps ITS 4 channels' nine taps (and bias) in
// registers.
// MODE 0: y = act(conv(x) + bias)          (forward)
// MODE 1: dz = da * gelu'(conv(x) + bias)  (backward prep: recomputes the pre-activation instead of saving it)
// MODE 2: dx (+)= conv^T(dy)               (data gradient: the same stencil with the taps mirrored)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void dw_stencil_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, const T* __restrict__ da,
                                                         T* __restrict__ out, RunGeom g, int C, int act, int accumulate) {
  const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cx) * 4;
  const long run = (long)blockIdx.y * 4 + py;
  if (c >= C || run >= g.nruns) return;
  float wr[9][4], bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 9; ++t) ld4(w + (MODE == 2 ? 8 - t : t) * C + c, wr[t]);
  if (MODE != 2 && bias) ld4(bias + c, bs);
  const unsigned ru = (unsigned)run;
  const int rr = (int)(ru % (unsigned)g.rpr);
  const unsigned bh = ru / (unsigned)g.rpr;  // b*H + h
  const int h = (int)(bh % (unsigned)g.H);
  const int rho = rr / g.rpc, k = rr - rho * g.rpc;
  const int w0 = rho + k * kRun * g.dil;
  if (w0 >= g.W) return;
  const long row_base = (long)bh * g.W;  // pixel index of (b, h, 0)
  const T* xc = x + c;
  float win[3][3][4];
  load_col(xc, row_base, h, w0 - g.dil, g.H, g.W, C, g.dil, win[0]);
  load_col(xc, row_base, h, w0, g.H, g.W, C, g.dil, win[1]);
#pragma unroll
  for (int i = 0; i < kRun; ++i) {
    const int wx = w0 + i * g.dil;
    if (wx >= g.W) break;
    load_col(xc, row_base, h, wx + g.dil, g.H, g.W, C, g.dil, win[(i + 2) % 3]);
    float acc[4] = {bs[0], bs[1], bs[2], bs[3]};
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += win[(i + kw) % 3][kh][j] * wr[kh * 3 + kw][j];
    const long pix = row_base + wx;
    T* o = out + pix * C + c;
    if (MODE == 0) {
      if (act == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = gelu_erf(acc[j]);
      }
    } else if (MODE == 1) {
      float gd[4];
      ld4(da + pix * C + c, gd);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = gd[j] * gelu_erf_grad(acc[j]);
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
// block = 64 channel-quads x 16 run lanes (1024 threads); every thread walks runs_per_block/16 runs of its quad with
// the same sliding window (4 loads per pixel).  The 16 run lanes fold into a 4-slot LDS array in four
// barrier-separated rounds (no LDS atomics), then ONE fp32 global atomic per (channel, tap) per block -- few, fat
// blocks keep the same-address atomic traffic low.
template <typename T>
__global__ __launch_bounds__(1024) void dw_bwd_weight_kernel(const T* __restrict__ dz, const T* __restrict__ x,
                                                             float* __restrict__ dw, float* __restrict__ dbias, RunGeom g,
                                                             int C, int runs_per_block) {
  __shared__ float red[4][64][41];
  const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;  // py = 0..15
  const int c = (blockIdx.x * 64 + cx) * 4;
  const long r0 = (long)blockIdx.y * runs_per_block;
  const long r1 = min(g.nruns, r0 + runs_per_block);
  float acc[9][4], accb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = 0.f;
  if (c < C) {
    const T* xc = x + c;
    for (long run = r0 + py; run < r1; run += 16) {
      const unsigned ru = (unsigned)run;
      const int rr = (int)(ru % (unsigned)g.rpr);
      const unsigned bh = ru / (unsigned)g.rpr;
      const int h = (int)(bh % (unsigned)g.H);
      const int rho = rr / g.rpc, k = rr - rho * g.rpc;
      const int w0 = rho + k * kRun * g.dil;
      if (w0 >= g.W) continue;
      const long row_base = (long)bh * g.W;
      float win[3][3][4];
      load_col(xc, row_base, h, w0 - g.dil, g.H, g.W, C, g.dil, win[0]);
      load_col(xc, row_base, h, w0, g.H, g.W, C, g.dil, win[1]);
#pragma unroll
      for (int i = 0; i < kRun; ++i) {
        const int wx = w0 + i * g.dil;
        if (wx >= g.W) break;
        load_col(xc, row_base, h, wx + g.dil, g.H, g.W, C, g.dil, win[(i + 2) % 3]);
        float gd[4];
        ld4(dz + (row_base + wx) * C + c, gd);
#pragma unroll
        for (int j = 0; j < 4; ++j) accb[j] += gd[j];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[kh * 3 + kw][j] += gd[j] * win[(i + kw) % 3][kh][j];
      }
    }
  }
  for (int round = 0; round < 4; ++round) {
    if ((py >> 2) == round) {
      float* slot = red[py & 3][cx];
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) slot[t * 4 + j] = (round ? slot[t * 4 + j] : 0.f) + acc[t][j];
#pragma unroll
      for (int j = 0; j < 4; ++j) slot[36 + j] = (round ? slot[36 + j] : 0.f) + accb[j];
    }
    __syncthreads();
  }
  for (int k = threadIdx.x; k < 64 * 40; k += blockDim.x) {
    const int gx = k / 40, v = k - gx * 40;
    const int cc = (blockIdx.x * 64 + gx) * 4;
    if (cc >= C) continue;
    const float s = red[0][gx][v] + red[1][gx][v] + red[2][gx][v] + red[3][gx][v];
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
  if ((C & 3) || dil < 1) return CMDA_ERR_SHAPE;
  const RunGeom g = run_geom(B, H, W, dil);
  if (too_big(npix) || too_big(g.nruns)) return CMDA_ERR_SHAPE;
  const int gx = (C / 4 + 63) / 64;
  dim3 grid(gx, (unsigned)((g.nruns + 3) / 4));
  CMDA_DISPATCH_DTYPE(dtype, CMDA_LAUNCH((dw_stencil_kernel<T, MODE>), grid, dim3(256), 0, stream, (const T*)x, w, bias,
                                         (const T*)da, (T*)out, g, C, act, accumulate));
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
  if ((C & 3) || dil < 1) return CMDA_ERR_SHAPE;
  const RunGeom g = run_geom(B, H, W, dil);
  if (too_big(npix) || too_big(g.nruns)) return CMDA_ERR_SHAPE;
  const int gx = (C / 4 + 63) / 64;
  int rpb = 128;  // runs per block (16 run lanes -> 8 runs per thread)
  while (rpb > 16 && (g.nruns + rpb - 1) / rpb * gx < 512) rpb >>= 1;
  dim3 grid(gx, (unsigned)((g.nruns + rpb - 1) / rpb));
  CMDA_DISPATCH_DTYPE(dtype, CMDA_LAUNCH((dw_bwd_weight_kernel<T>), grid, dim3(1024), 0, stream, (const T*)dz,
                                         (const T*)x, dw, dbias, g, C, rpb));
  CMDA_CHECK_LAUNCH();
}
