// hip_emu.h -- TEST INFRASTRUCTURE ONLY.
//
// A small host-side emulator of the HIP execution model (grid / block / 64-lane
// wavefront, LDS, __syncthreads, wave shuffles, MFMA tiles, ds_read_tr16) that
// lets the *same* kernel sources under cmda_amd/csrc/ be compiled with clang++
// for x86 and executed on CPU at small sizes.  It exists because the authoring
// container has no GPU: kernel indexing is checked here against the oracle
// before a GPU box is spent on it.  It is never loaded by the product path
// (cmda_amd/_lib.py only ever opens libcmda_hip.so and requires CUDA tensors).
//
// Model: every GPU thread of a block is a fiber (hand-rolled x86-64 context
// switch); the fibers of one block run on one OS thread; blocks are spread over
// a few OS worker threads.  Block- and wave-level collectives are rendezvous
// points that yield to the block scheduler.
#pragma once
#ifndef CMDA_EMU
#error "hip_emu.h is only for the CMDA_EMU host build"
#endif
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>
#include <sys/mman.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }

namespace emu {

constexpr int kWave = 64;
constexpr size_t kStack = 128 * 1024;

extern "C" void emu_switch(void** save_sp, void* load_sp);

struct WaveState {
  int active = 0;      // lanes not yet exited
  int arrived = 0;
  unsigned gen = 0;
  // exchange slots for collectives
  uint64_t slot64[kWave];
  float fa[kWave][8], fb[kWave][8];   // mfma operand staging (as float)
  const void* ptr[kWave];
};

struct BlockState {
  int nthreads = 0;
  int active = 0;      // threads not yet exited
  int arrived = 0;
  unsigned gen = 0;
  unsigned long progress = 0;
  std::vector<WaveState> waves;
};

struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  bool done = false;
  dim3 tid;
  int lin = 0;
};

struct Worker {
  std::vector<Fiber> fibers;
  void* sched_sp = nullptr;
  int cur = -1;
  BlockState blk;
  const std::function<void()>* body = nullptr;
  char* dyn_smem = nullptr;
  size_t dyn_smem_cap = 0;
};

extern thread_local Worker* g_worker;
extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

void yield_to_sched();
void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body);
void block_barrier();
void wave_barrier();
inline WaveState& my_wave() { return g_worker->blk.waves[g_worker->fibers[g_worker->cur].lin / kWave]; }
inline int my_lane() { return g_worker->fibers[g_worker->cur].lin % kWave; }
inline char* dyn_smem() { return g_worker->dyn_smem; }

}  // namespace emu

using emu::threadIdx;
using emu::blockIdx;
using emu::blockDim;
using emu::gridDim;

static inline void __syncthreads() { emu::block_barrier(); }
using std::min;
using std::max;

// ---- wave collectives (all lanes of the wave must call, convergent code) ----
template <typename T>
static inline T emu_shfl_idx(T v, int src_lane) {
  static_assert(sizeof(T) <= 8, "shfl payload");
  auto& w = emu::my_wave();
  int lane = emu::my_lane();
  uint64_t bits = 0;
  memcpy(&bits, &v, sizeof(T));
  w.slot64[lane] = bits;
  emu::wave_barrier();
  uint64_t r = w.slot64[src_lane & 63];
  emu::wave_barrier();
  T out;
  memcpy(&out, &r, sizeof(T));
  return out;
}
template <typename T> static inline T __shfl_xor(T v, int mask, int = 64) { return emu_shfl_idx(v, emu::my_lane() ^ mask); }
template <typename T> static inline T __shfl_down(T v, int d, int = 64) {
  int l = emu::my_lane();
  return emu_shfl_idx(v, l + d < 64 ? l + d : l);
}
template <typename T> static inline T __shfl(T v, int src, int = 64) { return emu_shfl_idx(v, src); }

// ---- atomics on "global" memory (host memory, blocks may run on several OS threads) ----
static inline float atomicAdd(float* p, float v) {
  uint32_t* u = reinterpret_cast<uint32_t*>(p);
  uint32_t old = __atomic_load_n(u, __ATOMIC_RELAXED);
  for (;;) {
    float f;
    memcpy(&f, &old, 4);
    float nf = f + v;
    uint32_t nu;
    memcpy(&nu, &nf, 4);
    if (__atomic_compare_exchange_n(u, &old, nu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return f;
  }
}
static inline double atomicAdd(double* p, double v) {
  uint64_t* u = reinterpret_cast<uint64_t*>(p);
  uint64_t old = __atomic_load_n(u, __ATOMIC_RELAXED);
  for (;;) {
    double f;
    memcpy(&f, &old, 8);
    double nf = f + v;
    uint64_t nu;
    memcpy(&nu, &nf, 8);
    if (__atomic_compare_exchange_n(u, &old, nu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return f;
  }
}
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) {
  return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}
static inline int atomicMax(int* p, int v) {
  int old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
static inline int atomicMin(int* p, int v) {
  int old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
static inline unsigned atomicMax(unsigned* p, unsigned v) {
  unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
static inline unsigned atomicMin(unsigned* p, unsigned v) {
  unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}

// ---- math spelled the HIP way ----
#define __expf(x) expf(x)
#define __logf(x) logf(x)
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __fdividef(float a, float b) { return a / b; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }

// ---- HIP vector types used by the kernels ----
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) ushort4 { unsigned short x, y, z, w; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }

// ---- launch ----
#define CMDA_LAUNCH(kernel, grid, block, smem, stream, ...) \
  emu::launch((grid), (block), (smem), [=]() { kernel(__VA_ARGS__); })
#define CMDA_DYN_SMEM(name) char* name = emu::dyn_smem()
