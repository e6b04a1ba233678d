// Probe: HBM write bandwidth of a tiled bf16 [M,N] output for different per-wave store footprints (gfx950).
//   mode 0: GEMM epilogue footprint — wave instruction = 8 rows x 128 B (row stride = N*2 bytes)
//   mode 1: 2 rows x 512 B per wave instruction (full tile-row segments)
//   mode 2: 1 KiB contiguous per wave instruction (row-major streaming, the fill pattern)
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/store_pattern.hip -o gpurun_out/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int NT_>
__global__ __launch_bounds__(512) void k(unsigned short* out, int M, int N) {
  const int ntn = N / 256;
  const int tm = blockIdx.x / ntn, tn = blockIdx.x % ntn;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const i32x4 v = {lane, w, tm, tn};
  if (MODE == 2) {  // flat: block covers 256*256 contiguous elements
    unsigned short* base = out + (size_t)blockIdx.x * 65536;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      i32x4* p = (i32x4*)(base + (it * 8 + w) * 512 + lane * 8);
      if (NT_) __builtin_nontemporal_store(v, p); else *p = v;
    }
    return;
  }
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    int row, col;
    if (MODE == 0) { row = (w >> 2) * 128 + it * 8 + (lane >> 3); col = (w & 3) * 64 + (lane & 7) * 8; }
    else { row = w * 32 + it * 2 + (lane >> 5); col = (lane & 31) * 8; }
    i32x4* p = (i32x4*)(out + (size_t)(tm * 256 + row) * N + tn * 256 + col);
    if (NT_) __builtin_nontemporal_store(v, p); else *p = v;
  }
}
// fp32 read-modify-write of a [M,N] fp32 matrix in the GEMM epilogue's direct footprint: wave instruction = 16 rows x 64 B
// (mode 0) vs 4 rows x 256 B (mode 1) — the residual-stream epilogue (out = resid + x).
template <int MODE>
__global__ __launch_bounds__(512) void rmw(const float* in, float* out, int M, int N) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int ntn = N / 256;
  const int tm = blockIdx.x / ntn, tn = blockIdx.x % ntn;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm = w >> 2, wn = w & 3;
#pragma unroll
  for (int g = 0; g < 8; ++g) {          // 8 groups of 4 instructions = 32 per wave (128 x 64 fp32 wave tile)
    f4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int row, col;
      if (MODE == 0) { row = wm * 128 + g * 16 + (lane & 15); col = wn * 64 + k * 16 + (lane >> 4) * 4; }
      else { row = wm * 128 + (g * 4 + k) * 4 + (lane >> 4); col = wn * 64 + (lane & 15) * 4; }
      v[k] = *(const f4*)(in + (size_t)(tm * 256 + row) * N + tn * 256 + col);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int row, col;
      if (MODE == 0) { row = wm * 128 + g * 16 + (lane & 15); col = wn * 64 + k * 16 + (lane >> 4) * 4; }
      else { row = wm * 128 + (g * 4 + k) * 4 + (lane >> 4); col = wn * 64 + (lane & 15) * 4; }
      __builtin_nontemporal_store(v[k] + 1.0f, (f4*)(out + (size_t)(tm * 256 + row) * N + tn * 256 + col));
    }
  }
}
template <int MODE> void run_rmw(float* a, float* b, int M, int N) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = (M / 256) * (N / 256);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((rmw<MODE>), dim3(grid), dim3(512), 0, 0, a, b, M, N);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((rmw<MODE>), dim3(grid), dim3(512), 0, 0, a, b, M, N);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("fp32 rmw N=%d mode %d: %7.1f us  %5.2f TB/s (read+write)\n", N, MODE, ms * 50, (double)M * N * 8 / (ms / 20 * 1e-3) / 1e12);
}
template <int MODE, int NT_> void run(unsigned short* d, int M, int N) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int grid = (M / 256) * (N / 256);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, NT_>), dim3(grid), dim3(512), 0, 0, d, M, N);
  hipEventRecord(a);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<MODE, NT_>), dim3(grid), dim3(512), 0, 0, d, M, N);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("N=%d mode %d nt=%d: %7.1f us  %5.2f TB/s\n", N, MODE, NT_, ms * 50, (double)M * N * 2 / (ms / 20 * 1e-3) / 1e12);
}
int main() {
  const int M = 32768;
  unsigned short* d; hipMalloc(&d, (size_t)M * 3072 * 2);
  for (int N : {768, 3072}) {
    run<0, 0>(d, M, N); run<0, 1>(d, M, N); run<1, 0>(d, M, N); run<1, 1>(d, M, N); run<2, 0>(d, M, N); run<2, 1>(d, M, N);
  }
  float *fa, *fb; hipMalloc(&fa, (size_t)M * 768 * 4); hipMalloc(&fb, (size_t)M * 768 * 4);
  run_rmw<0>(fa, fb, M, 768); run_rmw<1>(fa, fb, M, 768); run_rmw<0>(fa, fa, M, 768); run_rmw<1>(fa, fa, M, 768);
  return 0;
}
