// Store / copy bandwidth probe for gfx950: how wide must a lane's store be, and how fast is a pure write burst?
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/store_bw.hip -o build/store_bw ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <typename V>
__global__ void fill_kernel(V* __restrict__ dst, long n, V val) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = val;
}
template <typename V>
__global__ void copy_kernel(const V* __restrict__ src, V* __restrict__ dst, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}
template <typename V>
__global__ void read_kernel(const V* __restrict__ src, float* __restrict__ sink, long n) {
  float a = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    V v = src[i];
    a += reinterpret_cast<const float*>(&v)[0];
  }
  if (a == 123.456f) sink[0] = a;
}
// GEMM-epilogue-like: each block writes a 128-row x 256-byte tile of a matrix with `ld` bytes between rows
template <typename V>
__global__ void tile_store_kernel(char* __restrict__ dst, long ld, int tiles_n, V val) {
  const long m0 = (long)(blockIdx.x / tiles_n) * 128, n0 = (long)(blockIdx.x % tiles_n) * 256;
  constexpr int VPR = 256 / sizeof(V);
  for (int id = threadIdx.x; id < 128 * VPR; id += blockDim.x) {
    const int row = id / VPR, q = id % VPR;
    *reinterpret_cast<V*>(dst + (m0 + row) * ld + n0 + q * sizeof(V)) = val;
  }
}

template <typename F>
static float time_us(F f, int iters = 10) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / iters;
}

int main() {
  const long bytes = 512L << 20;
  char *src, *dst; float* sink;
  hipMalloc(&src, bytes); hipMalloc(&dst, bytes); hipMalloc(&sink, 4);
  hipMemset(src, 1, bytes); hipMemset(dst, 0, bytes);
  for (int grid : {2048, 8192, 32768}) {
    float t;
    t = time_us([&] { fill_kernel<float><<<grid, 256>>>((float*)dst, bytes / 4, 1.f); });
    printf("grid %6d fill  4B/lane  %8.1f us  %6.2f TB/s\n", grid, t, bytes / t / 1e6);
    t = time_us([&] { fill_kernel<float2><<<grid, 256>>>((float2*)dst, bytes / 8, make_float2(1.f, 2.f)); });
    printf("grid %6d fill  8B/lane  %8.1f us  %6.2f TB/s\n", grid, t, bytes / t / 1e6);
    t = time_us([&] { fill_kernel<float4><<<grid, 256>>>((float4*)dst, bytes / 16, make_float4(1.f, 2.f, 3.f, 4.f)); });
    printf("grid %6d fill 16B/lane  %8.1f us  %6.2f TB/s\n", grid, t, bytes / t / 1e6);
    t = time_us([&] { copy_kernel<float2><<<grid, 256>>>((const float2*)src, (float2*)dst, bytes / 8); });
    printf("grid %6d copy  8B/lane  %8.1f us  %6.2f TB/s (r+w)\n", grid, t, 2.0 * bytes / t / 1e6);
    t = time_us([&] { copy_kernel<float4><<<grid, 256>>>((const float4*)src, (float4*)dst, bytes / 16); });
    printf("grid %6d copy 16B/lane  %8.1f us  %6.2f TB/s (r+w)\n", grid, t, 2.0 * bytes / t / 1e6);
    t = time_us([&] { read_kernel<float2><<<grid, 256>>>((const float2*)src, sink, bytes / 8); });
    printf("grid %6d read  8B/lane  %8.1f us  %6.2f TB/s\n", grid, t, bytes / t / 1e6);
    t = time_us([&] { read_kernel<float4><<<grid, 256>>>((const float4*)src, sink, bytes / 16); });
    printf("grid %6d read 16B/lane  %8.1f us  %6.2f TB/s\n", grid, t, bytes / t / 1e6);
  }
  // 262144 x 2048-byte matrix (the hd dpw output), 128x256B tiles
  const long ld = 2048; const int tiles_n = 8; const int nt = (int)(bytes / ld / 128) * tiles_n;
  float t = time_us([&] { tile_store_kernel<float2><<<nt, 256>>>(dst, ld, tiles_n, make_float2(1.f, 2.f)); });
  printf("tile store  8B/lane (128 rows x 256 B, ld 2048)  %8.1f us  %6.2f TB/s\n", t, bytes / t / 1e6);
  t = time_us([&] { tile_store_kernel<float4><<<nt, 256>>>(dst, ld, tiles_n, make_float4(1.f, 2.f, 3.f, 4.f)); });
  printf("tile store 16B/lane (128 rows x 256 B, ld 2048)  %8.1f us  %6.2f TB/s\n", t, bytes / t / 1e6);
  return 0;
}
