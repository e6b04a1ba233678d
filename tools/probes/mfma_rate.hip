// Probe: sustained rate of the two bf16 MFMA shapes on gfx950 from independent accumulators (no LDS, no memory), 1..3 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_rate.hip -o tools/probes/mfma_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(1.0f + j * 0.01f); }
  float s = 0.f;
  if constexpr (SHAPE == 32) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int m = 0; m < 16; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % NACC], 0, 0, 0);
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int m = 0; m < 32; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m % NACC], 0, 0, 0);
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  }
  if (s == 1.2345e30f) out[0] = s;
}

template <int SHAPE, int NACC> void run(float* d, int w) {
  const int iters = 4000, blocks = 256 * w;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(blocks), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(blocks), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double sec = ms * 1e-3 / 5, flops = (double)blocks * 4 * iters * 16 * 32768.0;
  printf("%s  acc=%2d waves/SIMD=%d  %8.1f us  %6.0f TFLOP/s (%.2f of 2.5 PF)\n", SHAPE == 32 ? "32x32x16" : "16x16x32", NACC, w, sec * 1e6,
         flops / sec / 1e12, flops / sec / 2.5e15);
}
int main() {
  float* d; (void)hipMalloc(&d, 4096);
  for (int w = 1; w <= 3; ++w) {
    run<32, 4>(d, w); run<32, 8>(d, w);
    run<16, 4>(d, w); run<16, 8>(d, w); run<16, 32>(d, w);
  }
  return 0;
}
