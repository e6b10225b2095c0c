// ce_loss.hip -- fused "bilinear x-up-sample + per-pixel cross-entropy (+ top-1 accuracy)" and the teacher's
// fused "up-sample + softmax + max + confidence count" (pseudo-labels).
//
// Reference:
//   BaseDecodeHead(Fusion).losses   decode_heads/decode_head.py:588-606
//     resize(seg_logit -> label size)                         ops/wrappers.py:9-28
//     F.cross_entropy(reduction='none', ignore_index=255)     losses/cross_entropy_loss.py:21-26
//     * weight, .mean() over ALL B*H*W pixels (ignored ones included in the denominator)  losses/utils.py:60-69
//     accuracy(): 100 * #(argmax == label) / numel            losses/accuracy.py:40-50
//   DACS teacher: encode_decode's resize + softmax + max + ge(0.968)   segmentors/encoder_decoder.py:733-745,
//     uda/dacs.py:674-682,702-705
//
// The 19 x H x W up-sampled logits are never materialised (the reference makes three passes over that 40 MB
// tensor): logits stay NHWC fp32 at 1/4 resolution (L2 resident, 1.2 MB per sample) and each thread rebuilds the 19
// class scores of one full-resolution pixel on the fly.  HBM-bound; algorithmic bytes per sample:
// fwd  19*h*w*4 + H*W*(8 label + 4 weight + 4 lse out); bwd the same + 19*h*w*4 written.
// Backward is a deterministic gather over the 1/4-resolution grid (no atomics), one thread per (pixel, class).
#include <cstdlib>
#include "bilinear.h"

namespace {
constexpr int kMaxClasses = 32;

template <int NC_MAX>
static __device__ __forceinline__ void upsample_scores(const float* __restrict__ lg, int b, int h, int w, int nc,
                                                       const BilinTap& ty, const BilinTap& tx, float (&s)[NC_MAX]) {
  const float* p00 = lg + ((long)(b * h + ty.i0) * w + tx.i0) * nc;
  const float* p01 = lg + ((long)(b * h + ty.i0) * w + tx.i1) * nc;
  const float* p10 = lg + ((long)(b * h + ty.i1) * w + tx.i0) * nc;
  const float* p11 = lg + ((long)(b * h + ty.i1) * w + tx.i1) * nc;
#pragma unroll
  for (int c = 0; c < NC_MAX; ++c)
    if (c < nc) s[c] = bilin_mix(p00[c], p01[c], p10[c], p11[c], tx.l0, tx.l1, ty.l0, ty.l1);
}

// Tiled form of the per-pixel kernels: a block owns kTW x kTH full-resolution pixels of one image and first copies the low-resolution
// logits its pixels can touch (a (kTW/scale + 2) x (kTH/scale + 2) patch, 19 floats each: ~4 KB at scale 4) into LDS with coalesced
// loads; the 4 x 19 reads per pixel then come from LDS.  (Straight from global memory every wave-load touched ~16 different
// 76-byte-strided lines: the forward took 62 us for 10 MB, staged 37 us; the pseudo-label kernel 40 -> 23 us; 32-row tiles -- a quarter
// of the same-address atomics -- change nothing.)  Same bilin_mix on the same values: bit-identical scores.
constexpr int kTW = 64, kTH = 8, kPatchFloats = 6144;

struct ScoreTile {
  int X0, Y0, b, tx0, ty0, ncols;
  bool staged;
};

static __device__ __forceinline__ ScoreTile stage_patch(const float* __restrict__ lg, float* __restrict__ patch, int h, int w, int H,
                                                        int W, int nc, float sh, float sw) {
  ScoreTile t;
  const int tiles_x = (W + kTW - 1) / kTW, tiles_y = (H + kTH - 1) / kTH;
  const int bt = blockIdx.x;
  t.b = bt / (tiles_x * tiles_y);
  const int r = bt - t.b * tiles_x * tiles_y;
  t.Y0 = (r / tiles_x) * kTH;
  t.X0 = (r % tiles_x) * kTW;
  const int X1 = min(t.X0 + kTW, W) - 1, Y1 = min(t.Y0 + kTH, H) - 1;
  t.tx0 = bilin_tap(t.X0, w, W, sw).i0;
  t.ty0 = bilin_tap(t.Y0, h, H, sh).i0;
  const int tx1 = bilin_tap(X1, w, W, sw).i1, ty1 = bilin_tap(Y1, h, H, sh).i1;
  t.ncols = tx1 - t.tx0 + 1;
  const int nrows = ty1 - t.ty0 + 1, rowf = t.ncols * nc;
  t.staged = rowf * nrows <= kPatchFloats;
  if (t.staged) {
    for (int rr = 0; rr < nrows; ++rr) {
      const float* src = lg + ((long)(t.b * h + t.ty0 + rr) * w + t.tx0) * nc;
      for (int k = threadIdx.x; k < rowf; k += blockDim.x) patch[rr * rowf + k] = src[k];
    }
  }
  __syncthreads();
  return t;
}

template <int NC_MAX>
static __device__ __forceinline__ void tile_scores(const ScoreTile& t, const float* __restrict__ lg, const float* __restrict__ patch,
                                                   int h, int w, int nc, const BilinTap& ty, const BilinTap& tx, float (&s)[NC_MAX]) {
  if (!t.staged) {
    upsample_scores<NC_MAX>(lg, t.b, h, w, nc, ty, tx, s);
    return;
  }
  const float* p00 = patch + ((ty.i0 - t.ty0) * t.ncols + (tx.i0 - t.tx0)) * nc;
  const float* p01 = patch + ((ty.i0 - t.ty0) * t.ncols + (tx.i1 - t.tx0)) * nc;
  const float* p10 = patch + ((ty.i1 - t.ty0) * t.ncols + (tx.i0 - t.tx0)) * nc;
  const float* p11 = patch + ((ty.i1 - t.ty0) * t.ncols + (tx.i1 - t.tx0)) * nc;
#pragma unroll
  for (int c = 0; c < NC_MAX; ++c)
    if (c < nc) s[c] = bilin_mix(p00[c], p01[c], p10[c], p11[c], tx.l0, tx.l1, ty.l0, ty.l1);
}

// acc[0] += sum_pix weight*nll ; acc[1] += #correct ; lse[pix] saved for the backward
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, const long long* __restrict__ label,
                                                      const float* __restrict__ weight, float* __restrict__ lse_out,
                                                      float* __restrict__ acc, int B, int h, int w, int H, int W, int nc,
                                                      int ignore_index) {
  __shared__ float red[2][4];
  __shared__ float patch[kPatchFloats];
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  const ScoreTile t = stage_patch(logits, patch, h, w, H, W, nc, sh, sw);
  float lsum = 0.f, csum = 0.f;
  const int X = t.X0 + (threadIdx.x & (kTW - 1));
  for (int Y = t.Y0 + (threadIdx.x / kTW); Y < min(t.Y0 + kTH, H); Y += 256 / kTW) {
    if (X >= W) break;
    const long i = ((long)t.b * H + Y) * W + X;
    const BilinTap ty = bilin_tap(Y, h, H, sh), tx = bilin_tap(X, w, W, sw);
    float s[kMaxClasses];
    tile_scores<kMaxClasses>(t, logits, patch, h, w, nc, ty, tx, s);
    float mx = -INFINITY;
    int am = 0;
#pragma unroll
    for (int c = 0; c < kMaxClasses; ++c)
      if (c < nc && s[c] > mx) { mx = s[c]; am = c; }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxClasses; ++c)
      if (c < nc) se += __expf(s[c] - mx);
    const float lse = mx + __logf(se);
    if (lse_out) lse_out[i] = lse;
    const long long lab = label[i];
    if (lab != ignore_index && lab >= 0 && lab < nc) {
      float sl = 0.f;
#pragma unroll
      for (int c = 0; c < kMaxClasses; ++c)
        if (c == (int)lab) sl = s[c];
      const float wgt = weight ? weight[i] : 1.f;
      lsum += wgt * (lse - sl);
    }
    csum += ((long long)am == lab) ? 1.f : 0.f;
  }
  lsum = wave_sum(lsum);
  csum = wave_sum(csum);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { red[0][wid] = lsum; red[1][wid] = csum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(acc + 0, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(acc + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

// dlogits[b,y,x,c] = gscale * sum over full-res pixels (Y,X) touching (y,x) of
//                    wy*wx * weight[Y,X] * (softmax_c(Y,X) - [c == label])          (0 for ignored pixels)
// one thread per low-res pixel, all classes in registers.
__global__ void ce_bwd_kernel(const float* __restrict__ logits, const long long* __restrict__ label,
                              const float* __restrict__ weight, const float* __restrict__ lse,
                              const float* __restrict__ gscale_ptr, float gscale_mul, float* __restrict__ dlogits, int B,
                              int h, int w, int H, int W, int nc, int ignore_index) {
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  const float ish = (float)H / (float)h, isw = (float)W / (float)w;
  const float gscale = (gscale_ptr ? *gscale_ptr : 1.f) * gscale_mul;
  const long total = (long)B * h * w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const unsigned iu = (unsigned)i, tu = iu / (unsigned)w, bu = tu / (unsigned)h;   // (32-bit: B*h*w < 2^32, checked by the launcher)
    const int x = (int)(iu - tu * (unsigned)w), y = (int)(tu - bu * (unsigned)h), b = (int)bu;
    int Y0 = (int)floorf(((float)y - 0.5f) * ish - 0.5f) - 1, Y1 = (int)ceilf(((float)y + 1.5f) * ish - 0.5f) + 1;
    int X0 = (int)floorf(((float)x - 0.5f) * isw - 0.5f) - 1, X1 = (int)ceilf(((float)x + 1.5f) * isw - 0.5f) + 1;
    if (y == 0) Y0 = 0;
    if (x == 0) X0 = 0;
    if (y == h - 1) Y1 = H - 1;
    if (x == w - 1) X1 = W - 1;
    Y0 = max(Y0, 0); X0 = max(X0, 0);
    Y1 = min(Y1, H - 1); X1 = min(X1, W - 1);
    float g[kMaxClasses];
#pragma unroll
    for (int c = 0; c < kMaxClasses; ++c) g[c] = 0.f;
    for (int Y = Y0; Y <= Y1; ++Y) {
      const BilinTap ty = bilin_tap(Y, h, H, sh);
      const float wy = (ty.i0 == y ? ty.l0 : 0.f) + (ty.i1 == y ? ty.l1 : 0.f);
      if (wy == 0.f) continue;
      for (int X = X0; X <= X1; ++X) {
        const BilinTap tx = bilin_tap(X, w, W, sw);
        const float wx = (tx.i0 == x ? tx.l0 : 0.f) + (tx.i1 == x ? tx.l1 : 0.f);
        if (wx == 0.f) continue;
        const long pi = ((long)b * H + Y) * W + X;
        const long long lab = label[pi];
        if (lab == ignore_index || lab < 0 || lab >= nc) continue;
        const float coef = wy * wx * (weight ? weight[pi] : 1.f);
        if (coef == 0.f) continue;
        float s[kMaxClasses];
        upsample_scores<kMaxClasses>(logits, b, h, w, nc, ty, tx, s);
        const float l = lse[pi];
#pragma unroll
        for (int c = 0; c < kMaxClasses; ++c)
          if (c < nc) g[c] += coef * (__expf(s[c] - l) - (c == (int)lab ? 1.f : 0.f));
      }
    }
    float* o = dlogits + i * nc;
#pragma unroll
    for (int c = 0; c < kMaxClasses; ++c)
      if (c < nc) o[c] = gscale * g[c];
  }
}

// The same gather with one thread per (low-resolution pixel, class) -- the form that runs whenever the up-sampling factor is
// <= 6 (the path's is 4).  exp(s_c - lse) needs no other class (the forward saved lse), so nothing is shared through LDS and
// nothing is synchronised: 19 adjacent lanes read the same label / weight / lse (one broadcast load) and 19 consecutive
// logits (one coalesced line); per full-resolution row the thread keeps the 2 x 3 low-resolution logits its column can touch
// in registers and rebuilds each pixel's score with the forward's own bilin_mix (same rounding, so the softmax is the
// forward's).  B*h*w*nc threads (622 k at 2 x 128 x 128 x 19) instead of 1024 LDS-bound blocks.
constexpr int kMaxTapsX = 16;

static __device__ __forceinline__ void lo_hi_range(int y, int h, int H, float ish, int& Y0, int& Y1) {
  Y0 = (int)floorf(((float)y - 0.5f) * ish - 0.5f) - 1;
  Y1 = (int)ceilf(((float)y + 1.5f) * ish - 0.5f) + 1;
  if (y == 0) Y0 = 0;
  if (y == h - 1) Y1 = H - 1;
  Y0 = max(Y0, 0);
  Y1 = min(Y1, H - 1);
}

// The same window for an up-sampling factor f that is a power of two (H = f h; f = 0: lo_hi_range): every product of the source-index
// arithmetic is exact then, full-resolution row Y has taps {floor((Y + 0.5) / f - 0.5), +1} and low-resolution row y receives from
// exactly Y in [f y - f/2, f y + f + f/2 - 1] (the border rows from everything beyond them): no margin rows to build and skip.
static __device__ __forceinline__ void lo_hi_window(int y, int h, int H, float ish, int f, int& Y0, int& Y1) {
  if (f == 0) {
    lo_hi_range(y, h, H, ish, Y0, Y1);
    return;
  }
  Y0 = y == 0 ? 0 : f * y - (f >> 1);
  Y1 = y == h - 1 ? H - 1 : f * y + f + (f >> 1) - 1;
  Y0 = max(Y0, 0);
  Y1 = min(Y1, H - 1);
}

__global__ __launch_bounds__(256) void ce_bwd_gather_kernel(const float* __restrict__ logits, const long long* __restrict__ label,
                                                            const float* __restrict__ weight, const float* __restrict__ lse,
                                                            const float* __restrict__ gscale_ptr, float gscale_mul,
                                                            float* __restrict__ dlogits, int B, int h, int w, int H, int W,
                                                            int nc, int ignore_index) {
  const long total = (long)B * h * w * nc;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  const float ish = (float)H / (float)h, isw = (float)W / (float)w;
  const unsigned iu = (unsigned)i, pu = iu / (unsigned)nc, tu = pu / (unsigned)w, bu = tu / (unsigned)h;   // (32-bit: checked by the launcher)
  const int c = (int)(iu - pu * (unsigned)nc), x = (int)(pu - tu * (unsigned)w), y = (int)(tu - bu * (unsigned)h), b = (int)bu;
  int Y0, Y1, X0, X1;
  lo_hi_range(y, h, H, ish, Y0, Y1);
  lo_hi_range(x, w, W, isw, X0, X1);
  // the column taps of this thread's x, once: weight onto x and which of the three cached columns each tap reads
  float wxs[kMaxTapsX], lx0[kMaxTapsX], lx1[kMaxTapsX];
  unsigned sel0 = 0, sel1 = 0;  // bit k: tap 0 reads column x (else x-1); tap 1 reads column x (else x+1)
  const int nX = X1 - X0 + 1;
#pragma unroll
  for (int k = 0; k < kMaxTapsX; ++k) {
    wxs[k] = 0.f; lx0[k] = 0.f; lx1[k] = 0.f;
    if (k < nX) {
      const BilinTap tx = bilin_tap(X0 + k, w, W, sw);
      wxs[k] = (tx.i0 == x ? tx.l0 : 0.f) + (tx.i1 == x ? tx.l1 : 0.f);
      lx0[k] = tx.l0;
      lx1[k] = tx.l1;
      sel0 |= (tx.i0 == x ? 1u : 0u) << k;
      sel1 |= (tx.i1 == x ? 1u : 0u) << k;
    }
  }
  const int xm = max(x - 1, 0), xp = min(x + 1, w - 1);
  const float* Lb = logits + (long)b * h * w * nc + c;
  float g = 0.f;
  for (int Y = Y0; Y <= Y1; ++Y) {
    const BilinTap ty = bilin_tap(Y, h, H, sh);
    const float wy = (ty.i0 == y ? ty.l0 : 0.f) + (ty.i1 == y ? ty.l1 : 0.f);
    if (wy == 0.f) continue;
    const float* r0 = Lb + (long)ty.i0 * w * nc;
    const float* r1 = Lb + (long)ty.i1 * w * nc;
    const float a0 = r0[(long)xm * nc], a1 = r0[(long)x * nc], a2 = r0[(long)xp * nc];
    const float b0 = r1[(long)xm * nc], b1 = r1[(long)x * nc], b2 = r1[(long)xp * nc];
    const long rowp = ((long)b * H + Y) * W + X0;
#pragma unroll
    for (int k = 0; k < kMaxTapsX; ++k) {
      if (k >= nX || wxs[k] == 0.f) continue;
      const long long lab = label[rowp + k];
      if (lab == ignore_index || lab < 0 || lab >= nc) continue;
      const float coef = wy * wxs[k] * (weight ? weight[rowp + k] : 1.f);
      if (coef == 0.f) continue;
      const bool s0 = (sel0 >> k) & 1u, s1 = (sel1 >> k) & 1u;
      const float sc = bilin_mix(s0 ? a1 : a0, s1 ? a1 : a2, s0 ? b1 : b0, s1 ? b1 : b2, lx0[k], lx1[k], ty.l0, ty.l1);
      g += coef * (__expf(sc - lse[rowp + k]) - (c == (int)lab ? 1.f : 0.f));
    }
  }
  dlogits[i] = (gscale_ptr ? *gscale_ptr : 1.f) * gscale_mul * g;
}

// TILED gather (the form that runs at the path's sizes).  The thread-per-(pixel, class) kernel above rebuilds exp(s_c - lse) of every
// full-resolution pixel in EVERY low-resolution pixel's window: ~100 exponentials, label / weight / lse loads and bilinear mixes per
// output where each full-resolution pixel feeds only the 2 x 2 low-resolution pixels of its taps (94 us per call at 4 x 128 x 128
// x 19 up-sampled x 4, eight calls per UDA step on the decode head's single lane).  Here a block owns kTLW x kTLH low-resolution
// pixels of one image; the per-pixel gradient coef * (softmax_c - onehot_c) of the full-resolution rows the tile can receive from is
// built ONCE into LDS, eight rows at a time, by one thread per full-resolution pixel (forward's tile_scores: same scores, same
// rounding), and thread (row group, x, c) of the tile then sums its column window of each row from LDS and weights the sum onto the
// kTLH rows it owns (320 threads = two row groups of 8 x 19).  Deterministic; sums associate per row, then per row group.
constexpr int kTLW = 8, kTLH = 4, kStripRows = 8, kStripCols = 48, kStripFloats = kStripRows * kStripCols * 20;
constexpr int kLowPatchFloats = (kTLW + 6) * (kTLH + 6) * 20;   // up-sampling factors >= 2: the region's taps span at most tile + 6 low-resolution rows / columns

// NCT: the class count at compile time (19: the path's; every per-class loop unrolls, the LDS reads of a pixel are in flight
// together) or 0 = run-time nc
template <int NCT>
__global__ __launch_bounds__(320) void ce_bwd_tile_kernel(const float* __restrict__ logits, const long long* __restrict__ label,
                                                          const float* __restrict__ weight, const float* __restrict__ lse,
                                                          const float* __restrict__ gscale_ptr, float gscale_mul,
                                                          float* __restrict__ dlogits, int B, int h, int w, int H, int W,
                                                          int nc_rt, int ignore_index, int fy, int fx) {
  const int nc = NCT ? NCT : nc_rt;
  __shared__ float G[kStripFloats];          // [strip row][region column][class]
  __shared__ float patch[kLowPatchFloats];   // low-resolution logits the region's taps read
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  const float ish = (float)H / (float)h, isw = (float)W / (float)w;
  const int tiles_x = (w + kTLW - 1) / kTLW, tiles_y = (h + kTLH - 1) / kTLH;
  const int bt = blockIdx.x;
  const int b = bt / (tiles_x * tiles_y);
  const int r = bt - b * tiles_x * tiles_y;
  const int y0 = (r / tiles_x) * kTLH, x0 = (r % tiles_x) * kTLW;
  const int y1 = min(y0 + kTLH, h) - 1, x1 = min(x0 + kTLW, w) - 1;
  // full-resolution region the tile receives from (lo_hi_range of its first and last pixel: conservative by a pixel or two; exact for
  // power-of-two factors fy / fx)
  int YR0, YR1, XR0, XR1, tmp;
  lo_hi_window(y0, h, H, ish, fy, YR0, tmp);
  lo_hi_window(y1, h, H, ish, fy, tmp, YR1);
  lo_hi_window(x0, w, W, isw, fx, XR0, tmp);
  lo_hi_window(x1, w, W, isw, fx, tmp, XR1);
  const int ncols = XR1 - XR0 + 1;           // <= kStripCols (launcher)
  // low-resolution patch: the taps of the region's corners
  const int py0 = bilin_tap(YR0, h, H, sh).i0, py1 = bilin_tap(YR1, h, H, sh).i1;
  const int px0 = bilin_tap(XR0, w, W, sw).i0, px1 = bilin_tap(XR1, w, W, sw).i1;
  const int pcols = px1 - px0 + 1, prows = py1 - py0 + 1, prowf = pcols * nc;   // (prows * prowf <= kLowPatchFloats: launcher)
  for (int rr = 0; rr < prows; ++rr) {
    const float* src = logits + ((long)(b * h + py0 + rr) * w + px0) * nc;
    for (int k = threadIdx.x; k < prowf; k += blockDim.x) patch[rr * prowf + k] = src[k];
  }
  // gather role: thread = (row group, x, c) of the tile: blockDim / (kTLW * nc) groups share the strip's rows, kTLH outputs each
  const int per = kTLW * nc, ngrp = max(1, (int)blockDim.x / per), grp = threadIdx.x / per, tig = threadIdx.x - grp * per;
  const int gx = tig / nc, gc = tig - gx * nc;
  const bool gather = grp < ngrp && x0 + gx <= x1;
  const int x = x0 + gx;
  float wxs[kMaxTapsX];
  int X0 = 0, nX = 0;
  if (gather) {
    int X1;
    lo_hi_window(x, w, W, isw, fx, X0, X1);
    nX = X1 - X0 + 1;
  }
#pragma unroll
  for (int k = 0; k < kMaxTapsX; ++k) {
    wxs[k] = 0.f;
    if (k < nX) {
      const BilinTap tx = bilin_tap(X0 + k, w, W, sw);
      wxs[k] = (tx.i0 == x ? tx.l0 : 0.f) + (tx.i1 == x ? tx.l1 : 0.f);
    }
  }
  float acc[kTLH];
#pragma unroll
  for (int j = 0; j < kTLH; ++j) acc[j] = 0.f;
  __syncthreads();

  for (int YS = YR0; YS <= YR1; YS += kStripRows) {
    const int nrows = min(kStripRows, YR1 - YS + 1);
    // ---- per-pixel gradients of the strip: one thread per full-resolution pixel, all classes
    for (int pi = threadIdx.x; pi < nrows * ncols; pi += blockDim.x) {
      const int ry = pi / ncols, rx = pi - ry * ncols;
      const int Y = YS + ry, X = XR0 + rx;
      float* gp = G + (ry * kStripCols + rx) * nc;
      const long gi = ((long)b * H + Y) * W + X;
      const long long lab = label[gi];
      const bool live = lab != ignore_index && lab >= 0 && lab < nc;
      const float coef = live ? (weight ? weight[gi] : 1.f) : 0.f;
      if (coef == 0.f) {
#pragma unroll
        for (int c = 0; c < (NCT ? NCT : kMaxClasses); ++c)
          if (c < nc) gp[c] = 0.f;
        continue;
      }
      const BilinTap ty = bilin_tap(Y, h, H, sh), tx = bilin_tap(X, w, W, sw);
      const float* p00 = patch + ((ty.i0 - py0) * pcols + (tx.i0 - px0)) * nc;
      const float* p01 = patch + ((ty.i0 - py0) * pcols + (tx.i1 - px0)) * nc;
      const float* p10 = patch + ((ty.i1 - py0) * pcols + (tx.i0 - px0)) * nc;
      const float* p11 = patch + ((ty.i1 - py0) * pcols + (tx.i1 - px0)) * nc;
      const float l = lse[gi];
#pragma unroll
      for (int c = 0; c < (NCT ? NCT : kMaxClasses); ++c) {
        if (c < nc) {
          const float sc = bilin_mix(p00[c], p01[c], p10[c], p11[c], tx.l0, tx.l1, ty.l0, ty.l1);
          gp[c] = coef * (__expf(sc - l) - (c == (int)lab ? 1.f : 0.f));
        }
      }
    }
    __syncthreads();
    // ---- gather: the group's rows of the strip; per row the thread's column window is summed once, then weighted onto its rows
    if (gather) {
      for (int ry = grp; ry < nrows; ry += ngrp) {
        const BilinTap ty = bilin_tap(YS + ry, h, H, sh);
        const float* gr = G + (ry * kStripCols + (X0 - XR0)) * nc + gc;
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < kMaxTapsX; ++k)
          if (k < nX) t += wxs[k] * gr[k * nc];
#pragma unroll
        for (int j = 0; j < kTLH; ++j) {
          const int y = y0 + j;
          const float wy = (ty.i0 == y ? ty.l0 : 0.f) + (ty.i1 == y ? ty.l1 : 0.f);
          acc[j] += wy * t;
        }
      }
    }
    __syncthreads();
  }
  // fold the row groups (group 0 writes): partial sums through the strip buffer
  if (ngrp > 1) {
    if (gather && grp > 0) {
#pragma unroll
      for (int j = 0; j < kTLH; ++j) G[((grp - 1) * per + tig) * kTLH + j] = acc[j];
    }
    __syncthreads();
    if (gather && grp == 0) {
      for (int o = 1; o < ngrp; ++o)
#pragma unroll
        for (int j = 0; j < kTLH; ++j) acc[j] += G[((o - 1) * per + tig) * kTLH + j];
    }
  }
  if (!gather || grp != 0) return;
  const float gs = (gscale_ptr ? *gscale_ptr : 1.f) * gscale_mul;
#pragma unroll
  for (int j = 0; j < kTLH; ++j) {
    const int y = y0 + j;
    if (y <= y1) dlogits[(((long)b * h + y) * w + x) * nc + gc] = gs * acc[j];
  }
}

// label = first arg-max of the up-sampled scores; prob = 1 / sum exp(s - max); count += (prob >= thr)
__global__ __launch_bounds__(256) void pseudo_label_kernel(const float* __restrict__ logits, long long* __restrict__ label_out,
                                                            float* __restrict__ prob_out, int* __restrict__ count, int B, int h, int w,
                                                            int H, int W, int nc, float thr) {
  __shared__ int red[4];
  __shared__ float patch[kPatchFloats];
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  const ScoreTile t = stage_patch(logits, patch, h, w, H, W, nc, sh, sw);
  int cnt = 0;
  const int X = t.X0 + (threadIdx.x & (kTW - 1));
  for (int Y = t.Y0 + (threadIdx.x / kTW); Y < min(t.Y0 + kTH, H); Y += 256 / kTW) {
    if (X >= W) break;
    const long i = ((long)t.b * H + Y) * W + X;
    const BilinTap ty = bilin_tap(Y, h, H, sh), tx = bilin_tap(X, w, W, sw);
    float s[kMaxClasses];
    tile_scores<kMaxClasses>(t, logits, patch, h, w, nc, ty, tx, s);
    float mx = -INFINITY;
    int am = 0;
#pragma unroll
    for (int c = 0; c < kMaxClasses; ++c)
      if (c < nc && s[c] > mx) { mx = s[c]; am = c; }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxClasses; ++c)
      if (c < nc) se += __expf(s[c] - mx);
    const float prob = 1.f / se;
    label_out[i] = am;
    if (prob_out) prob_out[i] = prob;
    cnt += prob >= thr ? 1 : 0;
  }
  float cf = wave_sum((float)cnt);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) red[wid] = (int)cf;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(count, red[0] + red[1] + red[2] + red[3]);
}

// out[b,c,Y,X] (NCHW fp32) = bilinear up-sample of NHWC logits: encode_decode's resize (encoder_decoder.py:733-745)
__global__ void upsample_logits_nchw_kernel(const float* __restrict__ logits, float* __restrict__ out, int B, int h,
                                            int w, int H, int W, int nc) {
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  const long total = (long)B * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const unsigned iu = (unsigned)i, tu = iu / (unsigned)W, bu = tu / (unsigned)H;   // (32-bit: B*H*W < 2^32, checked by the launcher)
    const int X = (int)(iu - tu * (unsigned)W), Y = (int)(tu - bu * (unsigned)H), b = (int)bu;
    const BilinTap ty = bilin_tap(Y, h, H, sh), tx = bilin_tap(X, w, W, sw);
    float s[kMaxClasses];
    upsample_scores<kMaxClasses>(logits, b, h, w, nc, ty, tx, s);
#pragma unroll
    for (int c = 0; c < kMaxClasses; ++c)
      if (c < nc) out[(((long)b * nc + c) * H + Y) * W + X] = s[c];
  }
}

// weight[b,y,x] = (count / total) with the first `top` and last `bottom` rows zeroed (dacs.py:702-711)
__global__ void pseudo_weight_kernel(const int* __restrict__ count, float* __restrict__ weight, int B, int H, int W,
                                     int top, int bottom) {
  const long total = (long)B * H * W;
  const float v = (float)((double)(*count) / (double)total);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int Y = (int)((i / W) % H);
    weight[i] = (Y < top || Y >= H - bottom) ? 0.f : v;
  }
}

static inline int grid_for(long n) { return (int)std::max<long>(1, std::min<long>((n + 255) / 256, 4096)); }
}  // namespace

// logits: fp32 NHWC [B,h,w,nc]; label int64 [B,H,W]; weight fp32 [B,H,W] or NULL; lse_out fp32 [B,H,W];
// acc: fp32[2], ACCUMULATED (caller zeroes): acc[0] = sum w*nll, acc[1] = #(argmax == label).
extern "C" int cmda_ce_upsample_fwd(const float* logits, const int64_t* label, const float* weight, float* lse_out,
                                    float* acc, int B, int h, int w, int H, int W, int nc, int ignore_index,
                                    void* stream) {
  if ((long)B * H * W <= 0) return CMDA_OK;
  if (nc <= 0 || nc > kMaxClasses) return CMDA_ERR_SHAPE;
  const long tiles = (long)B * ((W + kTW - 1) / kTW) * ((H + kTH - 1) / kTH);
  if (tiles > 0x7fffffffL) return CMDA_ERR_SHAPE;
  CMDA_LAUNCH(ce_fwd_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, logits, (const long long*)label, weight, lse_out, acc, B, h,
              w, H, W, nc, ignore_index);
  CMDA_CHECK_LAUNCH();
}

// dlogits (fp32 NHWC [B,h,w,nc]) = (*gscale_ptr) * gscale_mul * d(sum w*nll)/dlogits; gscale_ptr may be NULL (=1).
extern "C" int cmda_ce_upsample_bwd(const float* logits, const int64_t* label, const float* weight, const float* lse,
                                    const float* gscale_ptr, float gscale_mul, float* dlogits, int B, int h, int w,
                                    int H, int W, int nc, int ignore_index, void* stream) {
  if ((long)B * h * w <= 0) return CMDA_OK;
  if (nc <= 0 || nc > kMaxClasses) return CMDA_ERR_SHAPE;
  if ((long)B * h * w * nc >= (1L << 32)) return CMDA_ERR_SHAPE;   // (32-bit index arithmetic in the kernels)
  // column taps one low-resolution pixel can receive (lo_hi_range's window): the register-resident form holds up to 16
  const long taps = (long)ceil(2.0 * W / w) + 4;
  // tiled form (90 -> 57 us at 2 x 128 x 128 x 19 -> 512 x 512, the step 58.4 -> 58.05 ms; CMDA_CE_GATHER=1: the kernel above, tuning A/B):
  // up-sampling factors >= 2 (low-resolution patch <= tile + 6 per axis), the tile's full-resolution region
  // ((kTLW + 1) * factor + 5 columns at most) inside the LDS strip, at most 20 classes
  {
    const long rcols = (long)ceil((kTLW + 1.0) * W / w) + 6;
    if (taps <= kMaxTapsX && 2L * h <= H && 2L * w <= W && rcols <= kStripCols && nc <= 20 && getenv("CMDA_CE_GATHER") == nullptr) {
      const long tiles = (long)B * ((w + kTLW - 1) / kTLW) * ((h + kTLH - 1) / kTLH);
      if (tiles > 0x7fffffffL) return CMDA_ERR_SHAPE;
      auto pow2_factor = [](int out, int in) { const int f = out / in; return (out % in) == 0 && f >= 2 && f <= 16 && (f & (f - 1)) == 0 ? f : 0; };
      const int fy = pow2_factor(H, h), fx = pow2_factor(W, w);
      if (nc == 19) CMDA_LAUNCH(ce_bwd_tile_kernel<19>, dim3((unsigned)tiles), dim3(320), 0, stream, logits, (const long long*)label, weight, lse, gscale_ptr,
                                gscale_mul, dlogits, B, h, w, H, W, nc, ignore_index, fy, fx);
      else CMDA_LAUNCH(ce_bwd_tile_kernel<0>, dim3((unsigned)tiles), dim3(320), 0, stream, logits, (const long long*)label, weight, lse, gscale_ptr,
                       gscale_mul, dlogits, B, h, w, H, W, nc, ignore_index, fy, fx);
      CMDA_CHECK_LAUNCH();
    }
  }
  if (taps <= kMaxTapsX && h <= H && w <= W) {
    const long blocks = ((long)B * h * w * nc + 255) / 256;
    if (blocks > 0x7fffffffL) return CMDA_ERR_SHAPE;
    CMDA_LAUNCH(ce_bwd_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, logits, (const long long*)label, weight, lse,
                gscale_ptr, gscale_mul, dlogits, B, h, w, H, W, nc, ignore_index);
    CMDA_CHECK_LAUNCH();
  }
  CMDA_LAUNCH(ce_bwd_kernel, dim3(grid_for((long)B * h * w)), dim3(256), 0, stream, logits, (const long long*)label,
              weight, lse, gscale_ptr, gscale_mul, dlogits, B, h, w, H, W, nc, ignore_index);
  CMDA_CHECK_LAUNCH();
}

// count: int32[1], ACCUMULATED (caller zeroes).
extern "C" int cmda_pseudo_label(const float* logits, int64_t* label_out, float* prob_out, int* count, int B, int h,
                                 int w, int H, int W, int nc, float thr, void* stream) {
  if ((long)B * H * W <= 0) return CMDA_OK;
  if (nc <= 0 || nc > kMaxClasses) return CMDA_ERR_SHAPE;
  const long tiles = (long)B * ((W + kTW - 1) / kTW) * ((H + kTH - 1) / kTH);
  if (tiles > 0x7fffffffL) return CMDA_ERR_SHAPE;
  CMDA_LAUNCH(pseudo_label_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, logits, (long long*)label_out, prob_out, count, B, h,
              w, H, W, nc, thr);
  CMDA_CHECK_LAUNCH();
}

extern "C" int cmda_pseudo_weight(const int* count, float* weight, int B, int H, int W, int top, int bottom,
                                  void* stream) {
  if ((long)B * H * W <= 0) return CMDA_OK;
  CMDA_LAUNCH(pseudo_weight_kernel, dim3(grid_for((long)B * H * W)), dim3(256), 0, stream, count, weight, B, H, W, top,
              bottom);
  CMDA_CHECK_LAUNCH();
}

extern "C" int cmda_upsample_logits_nchw(const float* logits, float* out, int B, int h, int w, int H, int W, int nc,
                                         void* stream) {
  if ((long)B * H * W <= 0) return CMDA_OK;
  if (nc <= 0 || nc > kMaxClasses) return CMDA_ERR_SHAPE;
  if ((long)B * H * W >= (1L << 32)) return CMDA_ERR_SHAPE;   // (32-bit index arithmetic in the kernel)
  CMDA_LAUNCH(upsample_logits_nchw_kernel, dim3(grid_for((long)B * H * W)), dim3(256), 0, stream, logits, out, B, h, w,
              H, W, nc);
  CMDA_CHECK_LAUNCH();
}
