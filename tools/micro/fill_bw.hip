// Per-CU operand fill rate probe for gfx950: how fast can ONE workgroup per CU pull a GEMM-like operand stream into LDS --
//   mode 0: LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, counted vmcnt, what gemm_kernels.h does),
//   mode 1: register staging (global_load_dwordx4 into VGPRs, ds_write_b128),
//   mode 2: global_load_dwordx4 alone (no LDS write: the load path's own ceiling)
// from (a) a per-workgroup 64 KiB region re-read every pass (L2-resident: 32 workgroups x 64 KiB per XCD), (b) a region shared by the
// 32 workgroups of an XCD (every line has 31 other readers: the GEMM case), (c) fresh memory every pass (HBM / Infinity Cache).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/fill_bw.hip -o build/fill_bw ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int kTile = 64 * 1024;   // bytes per pass per workgroup (one k-tile of a 256x256 bf16 tile)

static __device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base,
                                   16, 0, 0);
}

// NW waves; every pass moves kTile bytes: wave w, piece j covers bytes [(w * PPW + j) * 1024, +1024) of the pass's source
template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW, 1) void fill_kernel(const char* __restrict__ src, long pass_stride, long wg_stride, int passes,
                                                          float* __restrict__ sink, long row_stride = 0) {
  __shared__ __attribute__((aligned(1024))) char lds[2 * kTile];
  constexpr int PPW = kTile / 1024 / NW;   // pieces per wave per pass
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  // row_stride > 0: the GEMM operand shape -- a piece is 8 rows x 128 bytes, rows `row_stride` bytes apart (a [rows][K] matrix walked
  // along K by the passes)
  const char* base = row_stride > 0 ? src + (long)blockIdx.x * wg_stride + ((long)wid * PPW * 8 + (lane >> 3)) * row_stride + (lane & 7) * 16
                                    : src + (long)blockIdx.x * wg_stride + (long)wid * PPW * 1024 + lane * 16;
  const long jstep = row_stride > 0 ? 8 * row_stride : 1024;
  float acc = 0.f;
  for (int p = 0; p < passes; ++p) {
    const char* s = base + (long)p * pass_stride;
    char* dst = lds + (p & 1) * kTile + wid * PPW * 1024;
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < PPW; ++j) glds16(s + j * jstep, dst + j * 1024);
      // one pass in flight while the next is issued (two LDS buffers), as the 2-stage GEMM pipeline does
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
    } else {
      float4 v[PPW];
#pragma unroll
      for (int j = 0; j < PPW; ++j) v[j] = *reinterpret_cast<const float4*>(s + j * jstep);
      if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) *reinterpret_cast<float4*>(dst + j * 1024 + lane * 16) = v[j];
      } else {
#pragma unroll
        for (int j = 0; j < PPW; ++j) acc += v[j].x;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (acc == 123.456f || lds[threadIdx.x] == 77) sink[0] = acc;
}

template <typename F>
static float time_us(F f, int iters = 5) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / iters;
}

template <int MODE, int NW>
static void run(const char* name, const char* src, long pass_stride, long wg_stride, int passes, int grid, float* sink, long row_stride = 0) {
  const float t = time_us([&] { fill_kernel<MODE, NW><<<grid, 64 * NW>>>(src, pass_stride, wg_stride, passes, sink, row_stride); });
  const double bytes = (double)grid * passes * kTile;
  printf("  %-44s %2d waves  %9.1f us  %7.1f GB/s per workgroup  %6.2f TB/s chip\n", name, NW, t, bytes / grid / t / 1e3, bytes / t / 1e6);
}

int main() {
  const int grid = 256, passes = 256;
  const long big = (long)grid * passes * kTile;   // 4 GiB would be needed for fully fresh data: use 1 GiB and wrap inside the kernel stride
  char* src; float* sink;
  const long bytes = 1L << 30;
  hipMalloc(&src, bytes); hipMalloc(&sink, 4);
  hipMemset(src, 1, bytes);
  (void)big;
  printf("one workgroup per CU (%d workgroups), %d passes of 64 KiB each\n", grid, passes);
  printf("(a) private 64 KiB region per workgroup, re-read every pass (L2-resident)\n");
  run<0, 8>("LDS-DMA", src, 0, kTile, passes, grid, sink);
  run<0, 4>("LDS-DMA", src, 0, kTile, passes, grid, sink);
  run<1, 8>("global_load_dwordx4 + ds_write_b128", src, 0, kTile, passes, grid, sink);
  run<1, 4>("global_load_dwordx4 + ds_write_b128", src, 0, kTile, passes, grid, sink);
  run<2, 8>("global_load_dwordx4 only", src, 0, kTile, passes, grid, sink);
  printf("(b) ONE 64 KiB region per pass shared by all workgroups, 16 MiB walked (every line has 255 other readers)\n");
  run<0, 8>("LDS-DMA", src, kTile, 0, passes, grid, sink);
  run<1, 8>("global_load_dwordx4 + ds_write_b128", src, kTile, 0, passes, grid, sink);
  run<2, 8>("global_load_dwordx4 only", src, kTile, 0, passes, grid, sink);
  printf("(c) fresh memory every pass: workgroup w, pass p reads 64 KiB at (p * 256 + w) * 64 KiB of a 1 GiB buffer (16 passes)\n");
  run<0, 8>("LDS-DMA", src, (long)grid * kTile, kTile, 64, grid, sink);
  run<1, 8>("global_load_dwordx4 + ds_write_b128", src, (long)grid * kTile, kTile, 64, grid, sink);
  run<2, 8>("global_load_dwordx4 only", src, (long)grid * kTile, kTile, 64, grid, sink);
  printf("(d) as (a) with 64 workgroups only (a quarter of the CUs)\n");
  run<0, 8>("LDS-DMA", src, 0, kTile, passes, 64, sink);
  run<1, 8>("global_load_dwordx4 + ds_write_b128", src, 0, kTile, passes, 64, sink);
  printf("(e) GEMM operand shape: a pass = 512 rows x 128 bytes of a row-major matrix, next pass = the next 128 bytes of the same rows;\n"
         "    all workgroups of an XCD-sized group of 32 read the SAME rows (a shared B panel), groups 512 rows apart\n");
  for (long stride : {640L, 2048L, 2560L, 4096L, 16384L, 16384L + 128, 16384L + 256}) {
    char nm[64];
    snprintf(nm, sizeof nm, "LDS-DMA, row stride %ld B", stride);
    const int np = (int)(stride / 128 < 128 ? stride / 128 : 128);
    // workgroup w reads rows of group w % 8 (blockIdx round-robins over the XCDs): wg_stride applied per group below via a 0 stride
    // and the base offset folded into the row index would need a second parameter; keep it simple: ALL workgroups share the rows
    run<0, 8>(nm, src, 128, 0, np, grid, sink, stride);
  }
  for (long stride : {2048L, 16384L, 16384L + 128}) {
    char nm[64];
    snprintf(nm, sizeof nm, "load + ds_write, row stride %ld B", stride);
    run<1, 8>(nm, src, 128, 0, (int)(stride / 128 < 128 ? stride / 128 : 128), grid, sink, stride);
  }
  printf("(f) as (e) but every workgroup its own 512 rows (an A panel per tile row)\n");
  for (long stride : {2048L, 16384L, 16384L + 128}) {
    char nm[64];
    snprintf(nm, sizeof nm, "LDS-DMA, row stride %ld B", stride);
    run<0, 8>(nm, src, 128, 512 * stride > (1L << 21) ? (1L << 21) + 4096 : 512 * stride, (int)(stride / 128 < 128 ? stride / 128 : 128), grid, sink, stride);   // (<= 256 * 2 MiB + 8.5 MiB < 1 GiB)
  }
  return 0;
}
