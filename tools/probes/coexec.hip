// Probe: do matrix-core (MFMA) and VALU / transcendental instructions of the waves of one SIMD execute concurrently on gfx950?
// Each wave runs ITERS iterations of [NM independent 32x32x16 bf16 MFMAs] + [NE v_exp_f32 + NF v_fma_f32 on independent registers],
// interleaved in program order.  Reported: cycles per iteration per SIMD for MFMA only, VALU only and the mix, at 1..3 waves/SIMD.
// If the mix costs max(MFMA, VALU) the pipes overlap; if it costs the sum they do not.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/coexec.hip -o gpurun_out/coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NM, int NE, int NF>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(1.0f + j * 0.01f); }
  float e[16], f[16];
  for (int j = 0; j < 16; ++j) { e[j] = -0.001f * (threadIdx.x + j); f[j] = 0.5f + j; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {   // a quarter of the work per u, interleaved
#pragma unroll
      for (int m = 0; m < NM / 4; ++m) acc[(u + m) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[(u + m) & 3], 0, 0, 0);
#pragma unroll
      for (int x = 0; x < NE / 4; ++x) { const int j = (u * (NE / 4) + x) & 15; e[j] = __builtin_amdgcn_exp2f(e[j]); asm volatile("" : "+v"(e[j])); }
#pragma unroll
      for (int x = 0; x < NF / 4; ++x) { const int j = (u * (NF / 4) + x) & 15; f[j] = __builtin_fmaf(f[j], 0.999f, 0.001f); asm volatile("" : "+v"(f[j])); }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int j = 0; j < 16; ++j) s += e[j] + f[j];
  if (s == 1.2345e30f) out[0] = s;
}

template <int NM, int NE, int NF> void run(const char* name, float* d, int waves_per_simd) {
  const int iters = 2000, blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = one per SIMD; blocks per CU = waves/SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NM, NE, NF>), dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<NM, NE, NF>), dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 5;
  printf("%-28s waves/SIMD=%d  %8.1f us  %7.1f ns per iteration per wave-slot (x2.4 = cycles at 2.4 GHz: %6.0f)\n", name, waves_per_simd, us,
         us * 1e3 / iters, us * 1e3 / iters * 2.4);
}

int main() {
  float* d; hipMalloc(&d, 4096);
  for (int w = 1; w <= 3; ++w) {
    run<16, 0, 0>("16 MFMA", d, w);
    run<0, 32, 0>("32 exp", d, w);
    run<0, 0, 64>("64 fma", d, w);
    run<0, 32, 64>("32 exp + 64 fma", d, w);
    run<16, 32, 0>("16 MFMA + 32 exp", d, w);
    run<16, 0, 64>("16 MFMA + 64 fma", d, w);
    run<16, 32, 64>("16 MFMA + 32 exp + 64 fma", d, w);
  }
  return 0;
}
