// common.h -- shared device helpers for the CMDA gfx950 kernels.
//
// Target: AMD Instinct MI355X (gfx950 / CDNA4) only.  64-lane wavefronts, MFMA
// 16x16x32 bf16 and 16x16x4 f32 tiles, LDS transposed reads (ds_read_b64_tr_b16).
// The same sources also compile under tests/emu/hip_emu.h (CMDA_EMU) so that the
// indexing of every kernel can be exercised on CPU by the test-suite; that build
// is test infrastructure and is never linked into libcmda_hip.so.
#pragma once
// slots of the BatchNorm column-statistics workspace (batchnorm.hip; also filled by the GEMM epilogue, cmda_gemm_params_t.colstats)
#define CMDA_BN_SLOTS 32
#include <stdint.h>
#include <algorithm>

#ifdef CMDA_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#define CMDA_LAUNCH(kernel, grid, block, smem, stream, ...) \
  hipLaunchKernelGGL(kernel, (grid), (block), (smem), (hipStream_t)(stream), __VA_ARGS__)
#define CMDA_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#endif

#include "../../include/cmda_hip.h"  // C-ABI status codes, dtype tags, parameter structs

typedef unsigned short bf16_t;  // raw bf16 bits

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bfx8;
typedef __attribute__((ext_vector_type(4))) __bf16 bfx4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;

static __device__ __forceinline__ float bf2f(bf16_t b) {
  unsigned u = ((unsigned)b) << 16;
  return __uint_as_float(u);
}
// round-to-nearest-even; NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
static __device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(unsigned short, h);
}

template <typename T> struct Num;
template <> struct Num<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static constexpr int kChunk = 4;  // elements per 16-byte chunk
};
template <> struct Num<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
  static constexpr int kChunk = 8;
};

template <typename T> static __device__ __forceinline__ float ldf(const T* p) { return Num<T>::ld(p); }
template <typename T> static __device__ __forceinline__ void stf(T* p, float v) { Num<T>::st(p, v); }

// 4-wide vector access (16 B for f32, 8 B for bf16); pointer must be aligned accordingly.
static __device__ __forceinline__ void ld4(const float* p, float (&v)[4]) {
  float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
static __device__ __forceinline__ void ld4(const bf16_t* p, float (&v)[4]) {
  u16x4 t = *reinterpret_cast<const u16x4*>(p);
  v[0] = bf2f(t[0]); v[1] = bf2f(t[1]); v[2] = bf2f(t[2]); v[3] = bf2f(t[3]);
}
static __device__ __forceinline__ void st4(float* p, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
static __device__ __forceinline__ void st4(bf16_t* p, const float (&v)[4]) {
  u16x4 t;
  t[0] = f2bf(v[0]); t[1] = f2bf(v[1]); t[2] = f2bf(v[2]); t[3] = f2bf(v[3]);
  *reinterpret_cast<u16x4*>(p) = t;
}

// ---- wave (64 lanes) reductions ----
static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
static __device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
static __device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- MFMA wrappers ----
// 16x16x32 bf16: lane l holds A[row l&15][k = 8*(l>>4)+j], B[k = 8*(l>>4)+j][col l&15], j=0..7;
// C/D: col = l&15, row = 4*(l>>4)+reg.
// 16x16x4 f32: lane l holds A[row l&15][k = l>>4], B[k = l>>4][col l&15]; same C/D map.
#ifndef CMDA_EMU
static __device__ __forceinline__ f32x4 mfma_bf16_16x16x32(u16x8 a, u16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfx8, a), __builtin_bit_cast(bfx8, b), c, 0, 0, 0);
}
static __device__ __forceinline__ f32x4 mfma_f32_16x16x4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// Transposed LDS read: per 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3
// of a 4x16 block of 16-bit elements; lane i receives column i (element q = row q).  EXEC must be full.
static __device__ __forceinline__ u16x4 lds_read_tr16(const bf16_t* p) {
  bfx4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfx4*)(p));
  return __builtin_bit_cast(u16x4, t);
}
#else
static inline f32x4 mfma_bf16_16x16x32(u16x8 a, u16x8 b, f32x4 c) {
  auto& w = emu::my_wave();
  const int l = emu::my_lane();
  for (int j = 0; j < 8; ++j) { w.fa[l][j] = bf2f(a[j]); w.fb[l][j] = bf2f(b[j]); }
  emu::wave_barrier();
  const int col = l & 15;
  f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * (l >> 4) + r;
    float acc = d[r];
    for (int g = 0; g < 4; ++g)
      for (int j = 0; j < 8; ++j) acc = fmaf(w.fa[16 * g + row][j], w.fb[16 * g + col][j], acc);
    d[r] = acc;
  }
  emu::wave_barrier();
  return d;
}
static inline f32x4 mfma_f32_16x16x4(float a, float b, f32x4 c) {
  auto& w = emu::my_wave();
  const int l = emu::my_lane();
  w.fa[l][0] = a;
  w.fb[l][0] = b;
  emu::wave_barrier();
  const int col = l & 15;
  f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * (l >> 4) + r;
    float acc = d[r];
    for (int g = 0; g < 4; ++g) acc = fmaf(w.fa[16 * g + row][0], w.fb[16 * g + col][0], acc);
    d[r] = acc;
  }
  emu::wave_barrier();
  return d;
}
static inline u16x4 lds_read_tr16(const bf16_t* p) {
  auto& w = emu::my_wave();
  const int l = emu::my_lane();
  w.ptr[l] = p;
  emu::wave_barrier();
  const int g = l >> 4, i = l & 15;
  u16x4 out;
  for (int q = 0; q < 4; ++q) {
    const bf16_t* src = static_cast<const bf16_t*>(w.ptr[16 * g + 4 * q + (i >> 2)]);
    out[q] = src[i & 3];
  }
  emu::wave_barrier();
  return out;
}
#endif

// Asynchronous 16-byte global -> LDS copy (global_load_lds_dwordx4): the wave writes 1 KiB at `lds_wave_base`
// (wave-uniform), lane l landing at +16*l, from per-lane source addresses.  Completion is tracked by vmcnt; hipcc
// drains it (vmcnt(0)) in front of __syncthreads().
#ifndef CMDA_EMU
static __device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
#else
static inline void glds16(const void* gsrc, void* lds_wave_base) {
  memcpy(static_cast<char*>(lds_wave_base) + 16 * emu::my_lane(), gsrc, 16);
}
#endif

// (An inline-asm form of the same copy -- invisible to the compiler's wait-count bookkeeping: with the builtin hipcc puts
// `s_waitcnt vmcnt(0)` in front of LDS reads it cannot prove disjoint from a pending DMA destination -- was tried on the round-5
// fused-MixFFN experiment and removed with it: no kernel of the tree needs it)

// Barrier of a multi-stage LDS-DMA pipeline: wait until at most N of this wave's DMA loads are still in flight (the older
// ones -- the stage about to be read -- have landed), then s_barrier WITHOUT the vmcnt(0) drain __syncthreads() implies,
// so the younger stages stay in flight across the barrier.
#ifndef CMDA_EMU
template <int N>
static __device__ __forceinline__ void pipe_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}
#else
template <int N>
static inline void pipe_barrier() { __syncthreads(); }
#endif

// Workgroup barrier for LDS hand-offs INSIDE an LDS-DMA pipeline: waits for this wave's LDS traffic only (lgkmcnt), not for the
// vector-memory queue -- __syncthreads() would drain every DMA piece and global store in flight (vmcnt(0)).
#ifndef CMDA_EMU
static __device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#else
static inline void lds_barrier() { __syncthreads(); }
#endif

// Wave-local wait for this wave's own LDS-DMA loads (no barrier): at most N still in flight.
#ifndef CMDA_EMU
template <int N>
static __device__ __forceinline__ void dma_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
#else
template <int N>
static inline void dma_wait() { emu::wave_barrier(); }  // emulated lanes are fibers: rendezvous so every lane's copy is done
#endif

// erf by Abramowitz & Stegun 7.1.26 (|abs error| < 1.5e-7, i.e. fp32 round-off level): ~12 VALU ops instead of the
// ~40 of libm's erff -- the exact-erf GELU of the reference (nn.GELU, mix_transformer.py:26) stays well inside the parity
// bound while the MixFFN stencil kernels stop being VALU-bound.
#ifdef CMDA_EMU
static inline float fast_rcp(float x) { return 1.0f / x; }
#else
static __device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }  // v_rcp_f32 (1 ulp), not the ~10-op IEEE division
#endif
// returns erf(u) and leaves exp(-u^2) in e (the GELU derivative needs the same exponential for its pdf term)
static __device__ __forceinline__ float erf_as_e(float u, float& e) {
  const float ax = fabsf(u);
  const float t = fast_rcp(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  e = __expf(-ax * ax);
  const float r = 1.0f - poly * e;
  return u < 0.f ? -r : r;
}
static __device__ __forceinline__ float erf_as(float x) {
  float e;
  return erf_as_e(x, e);
}
static __device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }
static __device__ __forceinline__ float gelu_erf_grad(float x) {
  float e;  // exp(-(x / sqrt 2)^2) = exp(-x^2 / 2): also the Gaussian of the pdf term
  const float cdf = 0.5f * (1.0f + erf_as_e(x * 0.70710678118654752440f, e));
  return cdf + x * (0.39894228040143267794f * e);
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Zero-fill as a KERNEL (never hipMemsetAsync): inside a captured hipGraph a memset becomes a memset NODE, and on ROCm 7.2 those
// were observed to run out of order with the neighbouring kernel nodes of the same stream when the graph is replayed (BatchNorm
// / InstanceNorm statistics accumulated on top of a workspace that was cleared too late: flaky NaNs and wrong losses from the
// second replay on, tools/dbg/lanes_dbg.py).  A kernel node is ordered like every other kernel.  Every translation unit that
// needs it gets its own copy (anonymous namespace).
namespace {
__global__ void cmda_zero_words_kernel(unsigned* __restrict__ p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline void cmda_zero_async(void* p, size_t bytes, void* stream) {  // bytes % 4 == 0
  const long n = (long)(bytes / 4);
  if (n <= 0) return;
  const int grid = (int)(n + 255) / 256 > 1024 ? 1024 : (int)((n + 255) / 256);
  CMDA_LAUNCH(cmda_zero_words_kernel, dim3(grid), dim3(256), 0, stream, (unsigned*)p, n);
}
}  // namespace

#ifdef CMDA_EMU
#define CMDA_CHECK_LAUNCH() return CMDA_OK
#else
#define CMDA_CHECK_LAUNCH() return (hipGetLastError() == hipSuccess) ? CMDA_OK : CMDA_ERR_HIP
#endif

// dtype dispatch: body sees `T`
#define CMDA_DISPATCH_DTYPE(dtype, ...)                  \
  do {                                                   \
    if ((dtype) == CMDA_F32) { typedef float T; __VA_ARGS__; }      \
    else if ((dtype) == CMDA_BF16) { typedef bf16_t T; __VA_ARGS__; } \
    else return CMDA_ERR_DTYPE;                          \
  } while (0)
