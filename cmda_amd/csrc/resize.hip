// resize.hip -- bilinear resize (align_corners=False) of NHWC feature maps, fused with the channel-concat write.
//
// Reference: resize() + torch.cat in decode_heads/daformer_head.py:263-275 (embeds of the 4 pyramid levels resized
// to 1/4 resolution and concatenated into 1024 channels).  The forward writes straight into its channel slice
// [coff, coff+C) of the concat buffer (row pitch ldy), so the concat never exists as a separate copy.
// Backward is a deterministic gather (no atomics): each input pixel sums the output pixels whose taps touch it.
// HBM-bound: fwd ~ (IH*IW + OH*OW)*C*sizeof(T) per sample; lanes run along C (8/16-byte accesses).
#include "bilinear.h"

namespace {

template <typename T>
__global__ void resize_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int IH, int IW, int OH, int OW,
                                  int C, int ldy, int coff) {
  const int cg = C >> 2;
  const float sh = (float)IH / (float)OH, sw = (float)IW / (float)OW;
  const long total = (long)B * OH * OW * cg;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // (32-bit index arithmetic: 64-bit integer division is a ~70-instruction emulation on gfx950 and six of them per output
    // made this HBM-bound kernel VALU-bound; the launcher rejects tensors of 2^32 or more 4-channel groups)
    const unsigned iu = (unsigned)i, pu = iu / (unsigned)cg, ru = pu / (unsigned)OW, bu = ru / (unsigned)OH;
    const int c = (int)(iu - pu * (unsigned)cg) * 4, ox = (int)(pu - ru * (unsigned)OW), oy = (int)(ru - bu * (unsigned)OH), b = (int)bu;
    const BilinTap ty = bilin_tap(oy, IH, OH, sh), tx = bilin_tap(ox, IW, OW, sw);
    float v00[4], v01[4], v10[4], v11[4], o[4];
    const T* xb = x + (long)b * IH * IW * C + c;
    ld4(xb + ((long)ty.i0 * IW + tx.i0) * C, v00);
    ld4(xb + ((long)ty.i0 * IW + tx.i1) * C, v01);
    ld4(xb + ((long)ty.i1 * IW + tx.i0) * C, v10);
    ld4(xb + ((long)ty.i1 * IW + tx.i1) * C, v11);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = bilin_mix(v00[j], v01[j], v10[j], v11[j], tx.l0, tx.l1, ty.l0, ty.l1);
    st4(y + ((long)(b * OH + oy) * OW + ox) * ldy + coff + c, o);
  }
}

// dx[b,iy,ix,c] = sum over (oy,ox) of dy[b,oy,ox,coff+c] * wy(oy->iy) * wx(ox->ix)
// NX: columns of the candidate window held in registers (0: any ratio, the column weights recomputed per tap).  With NX > 0 the
// column weights of this thread's ix are computed ONCE and a row's NX loads are issued together: the x8 level (16 -> 128: an
// 18 x 18 window) spent its time re-deriving the same 18 bilinear taps for each of its 18 rows (126 -> 115 us with the loads
// batched alone).
template <typename T, int NX>
__global__ void resize_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int IH, int IW, int OH, int OW,
                                  int C, int ldy, int coff) {
  const int cg = C >> 2;
  const float sh = (float)IH / (float)OH, sw = (float)IW / (float)OW;
  const float ish = (float)OH / (float)IH, isw = (float)OW / (float)IW;
  const long total = (long)B * IH * IW * cg;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const unsigned iu = (unsigned)i, pu = iu / (unsigned)cg, ru = pu / (unsigned)IW, bu = ru / (unsigned)IH;   // (32-bit: see resize_fwd_kernel)
    const int c = (int)(iu - pu * (unsigned)cg) * 4, ix = (int)(pu - ru * (unsigned)IW), iy = (int)(ru - bu * (unsigned)IH), b = (int)bu;
    // conservative candidate window: src in [iy-1, iy+1)  <=>  oy in [(iy-0.5)*ish-0.5, (iy+1.5)*ish-0.5)
    int oy0 = (int)floorf(((float)iy - 0.5f) * ish - 0.5f) - 1, oy1 = (int)ceilf(((float)iy + 1.5f) * ish - 0.5f) + 1;
    int ox0 = (int)floorf(((float)ix - 0.5f) * isw - 0.5f) - 1, ox1 = (int)ceilf(((float)ix + 1.5f) * isw - 0.5f) + 1;
    if (iy == 0) oy0 = 0;          // clamped sources (src < 0 -> 0) all land on row 0
    if (ix == 0) ox0 = 0;
    if (iy == IH - 1) oy1 = OH - 1;
    if (ix == IW - 1) ox1 = OW - 1;
    oy0 = max(oy0, 0); ox0 = max(ox0, 0);
    oy1 = min(oy1, OH - 1); ox1 = min(ox1, OW - 1);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (NX > 0) {
      float wxs[NX];
      int oxs[NX];
#pragma unroll
      for (int u = 0; u < NX; ++u) {
        const int ox = ox0 + u;
        oxs[u] = min(ox, ox1);
        const BilinTap tx = bilin_tap(oxs[u], IW, OW, sw);
        wxs[u] = ox <= ox1 ? (tx.i0 == ix ? tx.l0 : 0.f) + (tx.i1 == ix ? tx.l1 : 0.f) : 0.f;
      }
      for (int oy = oy0; oy <= oy1; ++oy) {
        const BilinTap ty = bilin_tap(oy, IH, OH, sh);
        const float wy = (ty.i0 == iy ? ty.l0 : 0.f) + (ty.i1 == iy ? ty.l1 : 0.f);
        if (wy == 0.f) continue;
        const T* row = dy + ((long)(b * OH + oy) * OW) * ldy + coff + c;
        float g[NX][4];
#pragma unroll
        for (int u = 0; u < NX; ++u) ld4(row + (long)oxs[u] * ldy, g[u]);
#pragma unroll
        for (int u = 0; u < NX; ++u) {
          const float wgt = wy * wxs[u];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] += g[u][j] * wgt;
        }
      }
    } else {
      for (int oy = oy0; oy <= oy1; ++oy) {
        const BilinTap ty = bilin_tap(oy, IH, OH, sh);
        const float wy = (ty.i0 == iy ? ty.l0 : 0.f) + (ty.i1 == iy ? ty.l1 : 0.f);
        if (wy == 0.f) continue;
        for (int ox = ox0; ox <= ox1; ++ox) {
          const BilinTap tx = bilin_tap(ox, IW, OW, sw);
          const float wx = (tx.i0 == ix ? tx.l0 : 0.f) + (tx.i1 == ix ? tx.l1 : 0.f);
          if (wx == 0.f) continue;
          float g[4];
          ld4(dy + ((long)(b * OH + oy) * OW + ox) * ldy + coff + c, g);
          const float wgt = wy * wx;
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] += g[j] * wgt;
        }
      }
    }
    st4(dx + ((long)(b * IH + iy) * IW + ix) * C + c, acc);
  }
}

static inline int grid_for(long n) { return (int)std::max<long>(1, std::min<long>((n + 255) / 256, 8192)); }
}  // namespace

extern "C" int cmda_bilinear_fwd(const void* x, void* y, int B, int IH, int IW, int OH, int OW, int C, int ldy,
                                 int coff, int dtype, void* stream) {
  if ((long)B * OH * OW * C <= 0) return CMDA_OK;
  if ((C & 3) || (ldy & 3) || (coff & 3) || coff + C > ldy) return CMDA_ERR_SHAPE;
  if ((long)B * OH * OW * (C / 4) >= (1L << 32)) return CMDA_ERR_SHAPE;   // (32-bit index arithmetic in the kernel)
  CMDA_DISPATCH_DTYPE(dtype, CMDA_LAUNCH((resize_fwd_kernel<T>), dim3(grid_for((long)B * OH * OW * (C / 4))), dim3(256),
                                         0, stream, (const T*)x, (T*)y, B, IH, IW, OH, OW, C, ldy, coff));
  CMDA_CHECK_LAUNCH();
}

extern "C" int cmda_bilinear_bwd(const void* dy, void* dx, int B, int IH, int IW, int OH, int OW, int C, int ldy,
                                 int coff, int dtype, void* stream) {
  if ((long)B * IH * IW * C <= 0) return CMDA_OK;
  if ((C & 3) || (ldy & 3) || (coff & 3) || coff + C > ldy) return CMDA_ERR_SHAPE;
  // columns of the candidate window: ceil(2 * OW / IW) + 4 at most (see the kernel)
  const long nx = (long)ceil(2.0 * OW / IW) + 4;
  if ((long)B * IH * IW * (C / 4) >= (1L << 32)) return CMDA_ERR_SHAPE;   // (32-bit index arithmetic in the kernel)
  const dim3 grid(grid_for((long)B * IH * IW * (C / 4)));
#define CMDA_RESIZE_BWD(NXV) \
  CMDA_DISPATCH_DTYPE(dtype, CMDA_LAUNCH((resize_bwd_kernel<T, NXV>), grid, dim3(256), 0, stream, (const T*)dy, (T*)dx, B, IH, IW, OH, OW, C, ldy, coff))
  if (nx <= 8) CMDA_RESIZE_BWD(8);
  else if (nx <= 12) CMDA_RESIZE_BWD(12);
  else if (nx <= 20) CMDA_RESIZE_BWD(20);
  else CMDA_RESIZE_BWD(0);
#undef CMDA_RESIZE_BWD
  CMDA_CHECK_LAUNCH();
}
