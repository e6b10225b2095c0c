// dwconv.hip -- depthwise 3x3 convolution (optionally dilated) on NHWC activations.
//
// Reference ops replaced:
//   DWConv (3x3, pad 1, bias, groups=C) + GELU inside MixFFN   mix_transformer.py:37-44,443-455
//   depthwise half of DepthwiseSeparableConvModule (3x3, dilation 6/12/18, no bias) in the sep-ASPP
//   decode_heads/sep_aspp_head.py:18-27 (mmcv DepthwiseSeparableConvModule.depthwise_conv)
//
// HBM-bound stencils: a thread owns 4 adjacent channels and a RUN of 8 pixels of one image row spaced `dil` apart.  It
// first issues ALL 3 x (RUN+2) window loads (clamped addresses, zero-selected afterwards -- no branch between them, so
// ~15 KiB per wave is in flight), then slides a 3x3 register window along the run: 3.75 coalesced 8/16-byte loads per
// output (lanes run along C) instead of 9.  Algorithmic bytes per pixel-channel:
// fwd 2*sizeof(T); gelu-bwd-prep 3*sizeof(T); bwd-data 2*sizeof(T); bwd-weight 2*sizeof(T).
// Depthwise weights/bias stay fp32.  The stencil kernels read a tap-major [9][C] copy of the reference's [C,1,3,3]
// parameter (rt.wdw: one tiny permute per optimizer step) so that the 4 channel weights of a lane are one coalesced
// 16-byte load per tap; the weight-gradient kernel still accumulates in the parameter's own [C,9] layout.
#include "common.h"

namespace {

// Runs: a row's columns split into `dil` residue classes; class rho holds w = rho, rho+dil, ... and is cut into runs
// of `run` outputs.  run id -> (b, h, rho, k); first column w0 = rho + k*run*dil.
struct RunGeom {
  int H, W, dil, rpc, rpr;  // rpc: runs per residue class, rpr = dil * rpc runs per row
  long nruns;
};
static inline RunGeom run_geom(int B, int H, int W, int dil, int run) {
  RunGeom g;
  g.H = H; g.W = W; g.dil = dil;
  const int per_class = (W + dil - 1) / dil;
  g.rpc = (per_class + run - 1) / run;
  g.rpr = dil * g.rpc;
  g.nruns = (long)B * H * g.rpr;
  return g;
}

// raw (unconverted) 4-channel vectors: bf16 stays packed in 2 VGPRs while it waits in the prefetched window
template <typename T> struct Raw;
template <> struct Raw<float> {
  float4 v;
  __device__ __forceinline__ void load(const float* p, bool ok) {
    v = *reinterpret_cast<const float4*>(p);
    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __device__ __forceinline__ void unpack(float (&o)[4]) const { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
};
template <> struct Raw<bf16_t> {
  u16x4 v;
  __device__ __forceinline__ void load(const bf16_t* p, bool ok) {
    v = *reinterpret_cast<const u16x4*>(p);
    if (!ok) v = (u16x4)(0);
  }
  __device__ __forceinline__ void unpack(float (&o)[4]) const {
    o[0] = bf2f(v[0]); o[1] = bf2f(v[1]); o[2] = bf2f(v[2]); o[3] = bf2f(v[3]);
  }
};
template <typename T> struct RunLen { static constexpr int value = 8; };
template <> struct RunLen<float> { static constexpr int value = 4; };  // fp32 (parity mode): half the window registers

// XCD-aware block order.  Workgroups are dealt round-robin to the 8 XCDs, each with its own L2, and neighbouring image
// rows share two of their three input rows: with the natural order those neighbours sit on different XCDs and every XCD
// fetches its own copy (PMC: 3.8 bytes fetched per byte written).  The grid is 1-D and hardware block `lin` is mapped so
// that each XCD walks one contiguous band of (row-block, channel-group) pairs.
struct BlockXY {
  int bx;   // channel group
  long by;  // run block
};
static __device__ __forceinline__ BlockXY xcd_block(int gx) {
  const unsigned nb = gridDim.x, lin = blockIdx.x;
  const unsigned q = nb / 8, r = nb % 8, xcd = lin % 8, loc = lin / 8;
  const unsigned logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  BlockXY b;
  b.bx = (int)(logical % (unsigned)gx);
  b.by = (long)(logical / (unsigned)gx);
  return b;
}

struct RunPos {
  int h, w0;
  long row_base;  // pixel index of (b, h, 0)
};
static __device__ __forceinline__ bool decode_run(const RunGeom& g, long run, int runlen, RunPos& r) {
  const unsigned ru = (unsigned)run;
  const int rr = (int)(ru % (unsigned)g.rpr);
  const unsigned bh = ru / (unsigned)g.rpr;  // b*H + h
  r.h = (int)(bh % (unsigned)g.H);
  const int rho = rr / g.rpc, k = rr - rho * g.rpc;
  r.w0 = rho + k * runlen * g.dil;
  r.row_base = (long)bh * g.W;
  return r.w0 < g.W;
}

// the whole 3 x (RUN+2) window of a run: every load is unconditional (out-of-image taps read a clamped in-image address and
// are zeroed by a select), so the compiler issues them back to back
template <typename T, int RUN>
static __device__ __forceinline__ void load_window(const T* __restrict__ xc, const RunGeom& g, const RunPos& r, int C,
                                                   Raw<T> (&raw)[3][RUN + 2]) {
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int ih = r.h + (kh - 1) * g.dil;
    const bool okh = ih >= 0 && ih < g.H;
    const T* rowp = xc + (r.row_base + (okh ? (long)(kh - 1) * g.dil * g.W : 0L)) * C;
#pragma unroll
    for (int ci = 0; ci < RUN + 2; ++ci) {
      const int iw = r.w0 + (ci - 1) * g.dil;
      const bool ok = okh && iw >= 0 && iw < g.W;
      raw[kh][ci].load(rowp + (long)(ok ? iw : r.w0) * C, ok);
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
                                                         T* __restrict__ out, RunGeom g, int C, int act, int accumulate,
                                                         int gx) {
  constexpr int RUN = RunLen<T>::value;
  const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
  const BlockXY blk = xcd_block(gx);
  const int c = (blk.bx * 64 + cx) * 4;
  const long run = blk.by * 4 + py;
  RunPos r;
  if (c >= C || run >= g.nruns || !decode_run(g, run, RUN, r)) return;
  Raw<T> raw[3][RUN + 2], rda[RUN];
  load_window<T, RUN>(x + c, g, r, C, raw);
  if (MODE == 1) {
#pragma unroll
    for (int i = 0; i < RUN; ++i) {
      const int wx = r.w0 + i * g.dil;
      rda[i].load(da + (r.row_base + (wx < g.W ? wx : r.w0)) * C + c, true);
    }
  }
  float wr[9][4], bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 9; ++t) ld4(w + (MODE == 2 ? 8 - t : t) * C + c, wr[t]);
  if (MODE != 2 && bias) ld4(bias + c, bs);
  float win[3][3][4];  // [column slot][kh][channel]
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    raw[kh][0].unpack(win[0][kh]);
    raw[kh][1].unpack(win[1][kh]);
  }
#pragma unroll
  for (int i = 0; i < RUN; ++i) {
    const int wx = r.w0 + i * g.dil;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) raw[kh][i + 2].unpack(win[(i + 2) % 3][kh]);
    float acc[4] = {bs[0], bs[1], bs[2], bs[3]};
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += win[(i + kw) % 3][kh][j] * wr[kh * 3 + kw][j];
    if (wx < g.W) {
      T* o = out + (r.row_base + wx) * C + c;
      if (MODE == 0) {
        if (act == 2) {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = gelu_erf(acc[j]);
        }
      } else if (MODE == 1) {
        float gd[4];
        rda[i].unpack(gd);
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
}

// Dilated stencils (the sep-ASPP branches: dilation 6 / 12 / 18 on 128 x 128 maps).  With dilation d the three input rows of an
// output row lie d rows apart -- 1.5 ... 4.7 MB of activations between them at 1024 channels, so in dw_stencil_kernel's row-major
// order they come from beyond the XCD's 4 MB L2 three times (the 16-image forward moved ~2.1 GB through the fabric for 0.54 GB in
// + 0.54 GB out and ran at that limit: 330 us).  A dilated 3x3 convolution is d*d independent unit-stride convolutions on the
// sub-lattices (h mod d, w mod d); a thread here owns 4 channels x a run of DRUN sub-columns of ONE sub-lattice and walks DOWN its
// sub-rows with a rolling three-row register window, so every input row is requested once per run (the +-1 halo columns are the
// neighbouring run's, same workgroup or the next one on the same XCD).  Loads run three sub-rows ahead of the arithmetic.
// MODE 0: y = conv(x) [+ bias]; MODE 2: dx (+)= conv^T(dy) (mirrored taps).
constexpr int DRUN = 4;
struct DilGeom {
  int H, W, dil, rpc;   // rpc: runs per sub-row = ceil(ceil(W / dil) / DRUN)
  long nthreads;        // B * dil * dil * rpc
};
template <typename T, int MODE, int CQ>
static __device__ __forceinline__ void dw_dilated_walk(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                       T* __restrict__ out, const DilGeom& g, int C, int accumulate, int c, unsigned tu,
                                                       bool do_stats, int imgs_per_group, float (&cs)[4], float (&cq)[4], int& grp) {
  // t -> (b, rho_h, rho_w, k): k fastest, so the runs of one sub-row sit in the same / the next workgroup
  const int k = (int)(tu % (unsigned)g.rpc);
  const unsigned t1 = tu / (unsigned)g.rpc;
  const int rho_w = (int)(t1 % (unsigned)g.dil);
  const unsigned t2 = t1 / (unsigned)g.dil;
  const int rho_h = (int)(t2 % (unsigned)g.dil), b = (int)(t2 / (unsigned)g.dil);
  const int nr = (g.H - rho_h + g.dil - 1) / g.dil;      // sub-rows / sub-columns of this sub-lattice
  const int ncol = (g.W - rho_w + g.dil - 1) / g.dil;
  const int k0 = k * DRUN;
  if (nr <= 0 || k0 >= ncol) return;
  if (do_stats) grp = b / imgs_per_group;
  const long img = (long)b * g.H * g.W;
  // window column ci <-> sub-column k0 - 1 + ci; out-of-image columns read a clamped address and are zeroed by a select
  long coff[DRUN + 2];
  bool cok[DRUN + 2];
#pragma unroll
  for (int ci = 0; ci < DRUN + 2; ++ci) {
    const int kc = k0 - 1 + ci;
    cok[ci] = kc >= 0 && kc < ncol;
    coff[ci] = (long)(rho_w + (cok[ci] ? kc : k0) * g.dil) * C + c;
  }
  float wr[9][4], bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) ld4(w + (MODE == 2 ? 8 - tp : tp) * C + c, wr[tp]);
  if (MODE != 2 && bias) ld4(bias + c, bs);

  Raw<T> raw[3][DRUN + 2];       // sub-row r waits in raw[r % 3]
  Raw<T> praw[3][DRUN];          // accumulate mode: the previous dx values of sub-row r, requested with it
  float win[3][DRUN + 2][4];     // ... and is unpacked into win[r % 3]; sub-row -1 / nr = zero padding
  float pcur[DRUN][4];           // previous values of the sub-row being written
  const bool acc_mode = MODE == 2 && accumulate;
  // MODE 0 with `stats`: column sums / sums of squares of the outputs this thread stores (the BatchNorm behind the dilated depthwise
  // convolution of the sep-ASPP, sep_aspp_head.py:18-27), added to the BatchNorm workspace of the image's group when the walk is done
  // (no select on the loaded value here: it would pin the wait for the data right behind the request -- the padding columns are
  // zeroed when the row is unpacked)
  auto request = [&](int r, Raw<T> (&dst)[DRUN + 2], Raw<T> (&pdst)[DRUN]) {
    if (r < nr) {
      const long rb = (img + (long)(rho_h + r * g.dil) * g.W) * C;
#pragma unroll
      for (int ci = 0; ci < DRUN + 2; ++ci) dst[ci].load(x + rb + coff[ci], true);
      if (acc_mode) {
#pragma unroll
        for (int i = 0; i < DRUN; ++i) pdst[i].load(out + rb + coff[i + 1], true);
      }
    }
  };
  auto unpack = [&](int r, const Raw<T> (&src)[DRUN + 2], float (&dst)[DRUN + 2][4]) {
#pragma unroll
    for (int ci = 0; ci < DRUN + 2; ++ci) {
      if (r < nr && cok[ci]) src[ci].unpack(dst[ci]);
      else dst[ci][0] = dst[ci][1] = dst[ci][2] = dst[ci][3] = 0.f;
    }
  };
  auto unpack_prev = [&](int r, const Raw<T> (&src)[DRUN]) {
    if (acc_mode && r < nr) {
#pragma unroll
      for (int i = 0; i < DRUN; ++i) src[i].unpack(pcur[i]);
    }
  };
  request(0, raw[0], praw[0]);
  request(1, raw[1], praw[1]);
  request(2, raw[2], praw[2]);
#pragma unroll
  for (int ci = 0; ci < DRUN + 2; ++ci) win[2][ci][0] = win[2][ci][1] = win[2][ci][2] = win[2][ci][3] = 0.f;   // sub-row -1
#pragma unroll
  for (int i = 0; i < DRUN; ++i) pcur[i][0] = pcur[i][1] = pcur[i][2] = pcur[i][3] = 0.f;
  unpack(0, raw[0], win[0]);
  unpack_prev(0, praw[0]);
  request(3, raw[0], praw[0]);
  for (int j0 = 0; j0 < nr; j0 += 3) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {   // sub-row j = j0 + s lives in slot s
      const int j = j0 + s;
      if (j < nr) {   // (thread-private trip count: no cross-lane operation inside)
        unpack(j + 1, raw[(s + 1) % 3], win[(s + 1) % 3]);
        T* orow = out + (img + (long)(rho_h + j * g.dil) * g.W) * C;
#pragma unroll
        for (int i = 0; i < DRUN; ++i) {
          float acc[4] = {bs[0], bs[1], bs[2], bs[3]};
          if (acc_mode) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = pcur[i][q];
          }
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
              for (int q = 0; q < 4; ++q) acc[q] += win[(s + 2 + kh) % 3][i + kw][q] * wr[kh * 3 + kw][q];
          if (k0 + i < ncol) {
            st4(orow + coff[i + 1], acc);
            if (do_stats) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                cs[q] += acc[q];
                cq[q] += acc[q] * acc[q];
              }
            }
          }
        }
        unpack_prev(j + 1, praw[(s + 1) % 3]);
        request(j + 4, raw[(s + 1) % 3], praw[(s + 1) % 3]);
      }
    }
  }
}

template <typename T, int MODE, int CQ>
__global__ __launch_bounds__(256) void dw_dilated_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, T* __restrict__ out, DilGeom g, int C,
                                                         int act, int accumulate, int gx, float* __restrict__ stats,
                                                         int imgs_per_group) {
  const int cx = threadIdx.x % CQ, py = threadIdx.x / CQ;
  const BlockXY blk = xcd_block(gx);
  const int c = (blk.bx * CQ + cx) * 4;
  const long t = blk.by * (256 / CQ) + py;
  // MODE 0 with `stats` (the BatchNorm statistics of the outputs, below): the workgroup's run lanes meet in LDS at the end, so a
  // thread without work stays for the barrier instead of leaving
  const bool do_stats = MODE == 0 && stats != nullptr;
  float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
  int grp = -1;
  if (c < C && t < g.nthreads) dw_dilated_walk<T, MODE, CQ>(x, w, bias, out, g, C, accumulate, c, (unsigned)t, do_stats, imgs_per_group, cs, cq, grp);
  if (!do_stats) return;
  // one (image, sub-lattice, run) x 4 channels per thread; the 256 / CQ run lanes of a channel quad fold through LDS when they belong
  // to the same statistics group (always, except across an image-group boundary), then ONE thread per quad adds eight sums to one of
  // the 32 slots of the group's workspace (per-thread atomics -- 7 M per launch at the decode head's size -- cost 37 us of a 190-us kernel)
  constexpr int RL = 256 / CQ;
  __shared__ float red[RL][CQ][8];
  __shared__ int rgrp[RL];
  if (cx == 0) rgrp[py] = grp;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    red[py][cx][q] = cs[q];
    red[py][cx][4 + q] = cq[q];
  }
  __syncthreads();
  if (c >= C) return;
  bool uniform = true;   // (threads past the end of the problem carry group -1 and zero sums)
  int g0 = -1;
#pragma unroll
  for (int r = 0; r < RL; ++r) {
    const int gr = rgrp[r];
    if (gr >= 0) {
      if (g0 < 0) g0 = gr;
      else if (gr != g0) uniform = false;
    }
  }
  if (g0 < 0) return;
  const unsigned slot = (unsigned)blk.by & (CMDA_BN_SLOTS - 1);
  if (uniform) {
    if (py != 0) return;
    float* wsl = stats + ((long)g0 * (CMDA_BN_SLOTS + 1) + slot) * 2 * (long)C + c;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float a = 0.f, b2 = 0.f;
#pragma unroll
      for (int r = 0; r < RL; ++r) {
        a += red[r][cx][q];
        b2 += red[r][cx][4 + q];
      }
      atomicAdd(wsl + q, a);
      atomicAdd(wsl + C + q, b2);
    }
  } else if (grp >= 0) {
    float* wsl = stats + ((long)grp * (CMDA_BN_SLOTS + 1) + slot) * 2 * (long)C + c;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      atomicAdd(wsl + q, cs[q]);
      atomicAdd(wsl + C + q, cq[q]);
    }
  }
}

// dw[c,tap] += sum_pix dz[pix,c] * x[pix+tap,c];  dbias[c] += sum_pix dz[pix,c]
// block = 64 channel-quads x 4 run lanes (256 threads: at ~166 VGPRs three such blocks share a CU, a 512-thread block
// could only run alone); every thread walks runs_per_block/4 runs of its quad with the same prefetched sliding window
// (4.75 loads per pixel).  The 4 run lanes meet in a 4-slot LDS array (no LDS atomics), then ONE fp32 global atomic per
// (channel, tap) per block -- few, fat blocks keep the same-address atomic traffic low.
template <typename T, int CQ>
__global__ __launch_bounds__(256) void dw_bwd_weight_kernel(const T* __restrict__ dz, const T* __restrict__ x,
                                                            float* __restrict__ dw, float* __restrict__ dbias, RunGeom g,
                                                            int C, int runs_per_block, int gx_groups) {
  constexpr int RUN = RunLen<T>::value;
  constexpr int RL = 256 / CQ;   // run lanes (block shape: see dw_gelu_bwd_fused_kernel)
  __shared__ float red[RL][CQ][41];
  const int cx = threadIdx.x % CQ, py = threadIdx.x / CQ;
  const BlockXY blk = xcd_block(gx_groups);
  const int c = (blk.bx * CQ + cx) * 4;
  const long r0 = blk.by * runs_per_block;
  const long r1 = min(g.nruns, r0 + runs_per_block);
  float acc[9][4], accb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = 0.f;
  if (c < C) {
    for (long run = r0 + py; run < r1; run += RL) {
      RunPos r;
      if (!decode_run(g, run, RUN, r)) continue;
      Raw<T> raw[3][RUN + 2], rdz[RUN];
      load_window<T, RUN>(x + c, g, r, C, raw);
#pragma unroll
      for (int i = 0; i < RUN; ++i) {
        const int wx = r.w0 + i * g.dil;
        const bool ok = wx < g.W;
        rdz[i].load(dz + (r.row_base + (ok ? wx : r.w0)) * C + c, ok);
      }
      float win[3][3][4];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        raw[kh][0].unpack(win[0][kh]);
        raw[kh][1].unpack(win[1][kh]);
      }
#pragma unroll
      for (int i = 0; i < RUN; ++i) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) raw[kh][i + 2].unpack(win[(i + 2) % 3][kh]);
        float gd[4];
        rdz[i].unpack(gd);  // zero beyond the row end
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
  {
    float* slot = red[py][cx];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) slot[t * 4 + j] = acc[t][j];
#pragma unroll
    for (int j = 0; j < 4; ++j) slot[36 + j] = accb[j];
  }
  __syncthreads();
  for (int k = threadIdx.x; k < CQ * 40; k += blockDim.x) {
    const int gx = k / 40, v = k - gx * 40;
    const int cc = (blk.bx * CQ + gx) * 4;
    if (cc >= C) continue;
    float s = 0.f;
#pragma unroll
    for (int l = 0; l < RL; ++l) s += red[l][gx][v];
    if (v < 36) {
      const int t = v >> 2, j = v & 3;
      atomicAdd(dw + (cc + j) * 9 + t, s);
    } else if (dbias) {
      atomicAdd(dbias + cc + (v - 36), s);
    }
  }
}

// Fused MixFFN backward step: dz = da * gelu'(conv(x) + bias)  AND  dw[c,tap] += sum dz * x[pix+tap], dbias[c] += sum dz in
// ONE pass over x / da (the prep and weight-gradient kernels above each walk the same 3 x (RUN+2) window of x; fused, the
// window is loaded once, dz never has to be re-read and one launch per MixFFN backward disappears).  Same block shape and
// reduction as dw_bwd_weight_kernel; the weight gradient sees dz before its rounding to bf16.
// Block shape: CQ channel-quads x (256 / CQ) run lanes.  The cross-block reduction is one fp32 atomic per (channel, tap) per block,
// and device-scope atomics are served beyond the XCD's L2 (~50 ns apiece on one address, and each one a memory-side transaction):
// with 64 quads x 4 run lanes the stage-1 shape (C = 256: ONE channel group) needed 1024 blocks x 2560 atomics and spent 60 % of
// its time in them (tools/dbg/dw_var.py: 84.8 us, 32.0 without the atomics).  16 quads x 16 run lanes keeps the grid (4 x the
// channel groups) with 1/4 ... 1/8 of the blocks along the pixel axis, i.e. that much fewer atomics per address and in total.
template <typename T, int CQ>
__global__ __launch_bounds__(256, 2) void dw_gelu_bwd_fused_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, const T* __restrict__ da,
                                                                T* __restrict__ dz, float* __restrict__ dw,
                                                                float* __restrict__ dbias, RunGeom g, int C,
                                                                int runs_per_block, int gx_groups) {
  constexpr int RUN = 4;  // (8 as in the other kernels needs all 256 VGPRs in bf16: one wave per SIMD)
  constexpr int RL = 256 / CQ;   // run lanes
  __shared__ float red[RL][CQ][41];
  const int cx = threadIdx.x % CQ, py = threadIdx.x / CQ;
  const BlockXY blk = xcd_block(gx_groups);
  const int c = (blk.bx * CQ + cx) * 4;
  const long r0 = blk.by * runs_per_block;
  const long r1 = min(g.nruns, r0 + runs_per_block);
  float acc[9][4], accb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = 0.f;
  if (c < C) {
    float wr[9][4], bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) ld4(w + t * C + c, wr[t]);
    if (bias) ld4(bias + c, bs);
    for (long run = r0 + py; run < r1; run += RL) {
      RunPos r;
      if (!decode_run(g, run, RUN, r)) continue;
      Raw<T> raw[3][RUN + 2], rda[RUN];
      load_window<T, RUN>(x + c, g, r, C, raw);
#pragma unroll
      for (int i = 0; i < RUN; ++i) {
        const int wx = r.w0 + i * g.dil;
        const bool ok = wx < g.W;
        rda[i].load(da + (r.row_base + (ok ? wx : r.w0)) * C + c, ok);
      }
      float win[3][3][4];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        raw[kh][0].unpack(win[0][kh]);
        raw[kh][1].unpack(win[1][kh]);
      }
#pragma unroll
      for (int i = 0; i < RUN; ++i) {
        const int wx = r.w0 + i * g.dil;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) raw[kh][i + 2].unpack(win[(i + 2) % 3][kh]);
        float z[4] = {bs[0], bs[1], bs[2], bs[3]};
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int j = 0; j < 4; ++j) z[j] += win[(i + kw) % 3][kh][j] * wr[kh * 3 + kw][j];
        float gd[4];
        rda[i].unpack(gd);  // zero beyond the row end
#pragma unroll
        for (int j = 0; j < 4; ++j) gd[j] *= gelu_erf_grad(z[j]);
        if (wx < g.W) st4(dz + (r.row_base + wx) * C + c, gd);
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
  {
    float* slot = red[py][cx];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) slot[t * 4 + j] = acc[t][j];
#pragma unroll
    for (int j = 0; j < 4; ++j) slot[36 + j] = accb[j];
  }
  __syncthreads();
  for (int k = threadIdx.x; k < CQ * 40; k += blockDim.x) {
    const int gx = k / 40, v = k - gx * 40;
    const int cc = (blk.bx * CQ + gx) * 4;
    if (cc >= C) continue;
    float s = 0.f;
#pragma unroll
    for (int l = 0; l < RL; ++l) s += red[l][gx][v];
    if (v < 36) {
      const int t = v >> 2, j = v & 3;
      atomicAdd(dw + (cc + j) * 9 + t, s);
    } else if (dbias) {
      atomicAdd(dbias + cc + (v - 36), s);
    }
  }
}

static inline bool too_big(long n) { return n >= (1L << 32); }
static inline int run_len(int dtype) { return dtype == CMDA_BF16 ? RunLen<bf16_t>::value : RunLen<float>::value; }

template <int MODE>
static int launch_stencil(const void* x, const float* w, const float* bias, const void* da, void* out, int B, int H, int W,
                          int C, int dil, int act, int accumulate, int dtype, void* stream, float* stats = nullptr,
                          int imgs_per_group = 1) {
  const long npix = (long)B * H * W;
  if (npix * C <= 0) return CMDA_OK;
  if ((C & 3) || dil < 1) return CMDA_ERR_SHAPE;
  if (too_big(npix)) return CMDA_ERR_SHAPE;
  const int gx = (C / 4 + 63) / 64;
  if (dil >= 2 && MODE != 1 && (MODE == 2 || act == 0)) {   // sub-lattice walk (dw_dilated_kernel; no fused activation there)
    DilGeom dg;
    dg.H = H; dg.W = W; dg.dil = dil;
    dg.rpc = ((W + dil - 1) / dil + DRUN - 1) / DRUN;
    dg.nthreads = (long)B * dil * dil * dg.rpc;
    constexpr int cq = 64;   // channel quads per workgroup (128 / 256 measured the same: tools/dbg/dw_var.py)
    const long nb = (dg.nthreads + 256 / cq - 1) / (256 / cq) * gx;
    if (too_big(dg.nthreads) || nb > 0x7fffffffL) return CMDA_ERR_SHAPE;
    CMDA_DISPATCH_DTYPE(dtype, CMDA_LAUNCH((dw_dilated_kernel<T, MODE == 1 ? 0 : MODE, cq>), dim3((unsigned)nb), dim3(256), 0, stream,
                                           (const T*)x, w, bias, (T*)out, dg, C, act, accumulate, gx, stats, imgs_per_group));
    CMDA_CHECK_LAUNCH();
  }
  if (stats) return CMDA_ERR_UNSUPPORTED;   // the fused statistics exist on the dilated walk only
  const RunGeom g = run_geom(B, H, W, dil, run_len(dtype));
  if (too_big(g.nruns)) return CMDA_ERR_SHAPE;
  const long nblk = (g.nruns + 3) / 4 * gx;
  if (nblk > 0x7fffffffL) return CMDA_ERR_SHAPE;
  dim3 grid((unsigned)nblk);
  CMDA_DISPATCH_DTYPE(dtype, CMDA_LAUNCH((dw_stencil_kernel<T, MODE>), grid, dim3(256), 0, stream, (const T*)x, w, bias,
                                         (const T*)da, (T*)out, g, C, act, accumulate, gx));
  CMDA_CHECK_LAUNCH();
}
}  // namespace

extern "C" int cmda_dwconv3x3_fwd(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C,
                                  int dil, int act, int dtype, void* stream) {
  return launch_stencil<0>(x, w, bias, nullptr, y, B, H, W, C, dil, act, 0, dtype, stream);
}

// ... with the column statistics of y (per group of imgs_per_group consecutive images) accumulated into the BatchNorm workspace
// `colstats` (groups x cmda_bn_ws_floats(C) floats, zero on entry): cmda_gemm_params_t.colstats for the depthwise producer.  dil >= 2.
extern "C" int cmda_dwconv3x3_fwd_stats(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C,
                                        int dil, float* colstats, int imgs_per_group, int dtype, void* stream) {
  if (!colstats || imgs_per_group < 1 || B % imgs_per_group || dil < 2) return CMDA_ERR_UNSUPPORTED;
  return launch_stencil<0>(x, w, bias, nullptr, y, B, H, W, C, dil, 0, 0, dtype, stream, colstats, imgs_per_group);
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
  const RunGeom g = run_geom(B, H, W, dil, run_len(dtype));
  if (too_big(npix) || too_big(g.nruns)) return CMDA_ERR_SHAPE;
  constexpr int cq = 16, rl = 256 / cq;   // block shape: see dw_gelu_bwd_fused_kernel
  const int gx = (C / 4 + cq - 1) / cq;
  // runs per run lane: 8 measured best at the decode head's shapes (16 x 128 x 128 x 1024, dilation 6 / 18: 307 us against 331 for
  // 16 and 362 for the former 64-quad x 4-lane blocks), fewer when the grid would not fill the chip
  int k = 8;
  while (k > 2 && (g.nruns + rl * k - 1) / (rl * k) * gx < 768) k >>= 1;
  const int rpb = rl * k;
  dim3 grid((unsigned)((g.nruns + rpb - 1) / rpb * gx));
  CMDA_DISPATCH_DTYPE(dtype, CMDA_LAUNCH((dw_bwd_weight_kernel<T, cq>), grid, dim3(256), 0, stream, (const T*)dz,
                                         (const T*)x, dw, dbias, g, C, rpb, gx));
  CMDA_CHECK_LAUNCH();
}

// dz = da * gelu'(conv(x) + bias); dw [C][9] += sum dz * x(tap); dbias [C] += sum dz   (dw / dbias ACCUMULATED; w tap-major [9][C])
extern "C" int cmda_dwconv3x3_gelu_bwd_fused(const void* x, const float* w, const float* bias, const void* da, void* dz,
                                             float* dw, float* dbias, int B, int H, int W, int C, int dil, int dtype,
                                             void* stream) {
  const long npix = (long)B * H * W;
  if (npix * C <= 0) return CMDA_OK;
  if ((C & 3) || dil < 1) return CMDA_ERR_SHAPE;
  const RunGeom g = run_geom(B, H, W, dil, 4);
  if (too_big(npix) || too_big(g.nruns)) return CMDA_ERR_SHAPE;
  constexpr int cq = 16, rl = 256 / cq;
  const int gx = (C / 4 + cq - 1) / cq;
  // runs per run lane: fat blocks (few atomics) as long as the grid still fills the chip about twice (tools/dbg/dw_var.py)
  int k = 16;
  while (k > 2 && (g.nruns + rl * k - 1) / (rl * k) * gx < 512) k >>= 1;
  const int rpb = rl * k;
  dim3 grid((unsigned)((g.nruns + rpb - 1) / rpb * gx));
  CMDA_DISPATCH_DTYPE(dtype, CMDA_LAUNCH((dw_gelu_bwd_fused_kernel<T, cq>), grid, dim3(256), 0, stream, (const T*)x, w, bias,
                                         (const T*)da, (T*)dz, dw, dbias, g, C, rpb, gx));
  CMDA_CHECK_LAUNCH();
}
