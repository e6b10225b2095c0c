// Register-resident MFMA peak probe for gfx950 (BASELINE.md section 4 / SURVEY.md 8d: "measure the bf16 MFMA peak with a
// register-resident loop; use the measured number as the roofline denominator").
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o build/mfma_peak ; run on the GPU box.
// Each wave keeps NACC independent accumulator tiles in VGPRs and issues back-to-back MFMAs on constant operands: no LDS,
// no memory traffic, so the rate is the matrix pipes' own ceiling at the clock the chip sustains under this load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int NACC>
__global__ __launch_bounds__(256) void mfma16_kernel(float* sink, int iters, float seed) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 1e-3f); b[i] = (__bf16)(seed - i * 1e-3f); }
  f32x4 acc[NACC];
  for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  if (s == 12345.678f) sink[0] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma32_kernel(float* sink, int iters, float seed) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 1e-3f); b[i] = (__bf16)(seed - i * 1e-3f); }
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j)
    for (int k = 0; k < 16; ++k) acc[j][k] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j)
    for (int k = 0; k < 16; ++k) s += acc[j][k];
  if (s == 12345.678f) sink[0] = s;
}

template <typename F>
static float time_ms(F f, int reps = 5) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  float* sink; hipMalloc(&sink, 4);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, clock %d MHz\n", prop.gcnArchName, cus, prop.clockRate / 1000);
  const int iters = 20000;
  for (int wpc : {4, 8}) {            // waves per CU (4 SIMDs): 1 or 2 waves per SIMD
    const int blocks = cus * wpc / 4;  // 256-thread blocks = 4 waves
    {
      constexpr int NACC = 8;
      float ms = time_ms([&] { mfma16_kernel<NACC><<<blocks, 256>>>(sink, iters, 0.5f); });
      double flops = 2.0 * 16 * 16 * 32 * NACC * (double)iters * blocks * 4;
      printf("mfma_f32_16x16x32_bf16  %d waves/CU, %d acc tiles: %8.3f ms  %8.1f TFLOP/s\n", wpc, NACC, ms, flops / ms / 1e9);
    }
    {
      constexpr int NACC = 4;
      float ms = time_ms([&] { mfma32_kernel<NACC><<<blocks, 256>>>(sink, iters, 0.5f); });
      double flops = 2.0 * 32 * 32 * 16 * NACC * (double)iters * blocks * 4;
      printf("mfma_f32_32x32x16_bf16  %d waves/CU, %d acc tiles: %8.3f ms  %8.1f TFLOP/s\n", wpc, NACC, ms, flops / ms / 1e9);
    }
  }
  return 0;
}
