// Probe: (1) per-CU store bandwidth as a function of how many CUs store at once, (2) whether a wave that has issued a
// GEMM-epilogue's worth of stores (16 / 32 x 1 KiB) can go on issuing MFMAs while they drain, or stalls at store issue.
//   mode 0: stores only        (R rounds x S store instructions per wave, GEMM epilogue footprint: 8 rows x 128 B, row stride LD)
//   mode 1: MFMAs only         (R rounds x NM 16x16x32 bf16 MFMAs per wave, 32 accumulator tiles, operands in registers)
//   mode 2: per round S stores, then NM MFMAs, no wait in between (the stores may drain under the MFMAs)
//   mode 3: per round S stores, s_waitcnt vmcnt(0), then NM MFMAs (fully serialised reference)
// One 512-thread block per CU (128 KiB dynamic LDS forces it); G blocks.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/store_overlap.hip -o tools/probes/store_overlap.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NT_>
__global__ __launch_bounds__(512) void k(unsigned short* out, int ld, int R, int S, int NM, float* sink) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  f32x4 acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a[4], b[8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) a[i][j] = (__bf16)(0.01f * ((lane * 7 + i * 3 + j) % 13 - 6));
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(0.02f * ((lane * 5 + i * 11 + j) % 11 - 5));
  // block's region: R tiles of 256 rows x 256 columns, laid out as rows of ld elements (tile r at column 256*(r % (ld/256)), row block ...)
  const int tpr = ld / 256;
  for (int r = 0; r < R; ++r) {
    if (MODE != 1) {
      const size_t tile = (size_t)blockIdx.x * R + r;
      unsigned short* base = out + (tile / tpr) * 256 * (size_t)ld + (tile % tpr) * 256;
#pragma unroll 4
      for (int it = 0; it < S; ++it) {
        const int row = (w >> 2) * 128 + (it & 15) * 8 + (lane >> 3), col = (w & 3) * 64 + (lane & 7) * 8;
        const i32x4 v = {lane, w, r, it};
        i32x4* p = (i32x4*)(base + (size_t)row * ld + col);
        if (NT_) __builtin_nontemporal_store(v, p); else *p = v;
      }
      if (MODE == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (MODE != 0) {
      for (int m = 0; m < NM; m += 32) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j)
            acc[i * 8 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i * 8 + j], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 1.2345e30f) sink[0] = s;
}

template <int MODE, int NT_>
float run(unsigned short* out, int ld, int G, int R, int S, int NM, float* sink) {
  auto kern = k<MODE, NT_>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(G), dim3(512), 131072, 0, out, ld, R, S, NM, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best * 1000.f;
}

int main() {
  const int ld = 3072, R = 24;
  const size_t elems = (size_t)256 * R * 65536;   // 256 blocks x R tiles x 64K elements
  unsigned short* out; float* sink;
  hipMalloc(&out, elems * 2 + (1 << 20)); hipMalloc(&sink, 4);
  hipMemset(out, 0, elems * 2);
  printf("# part 1: stores only, S=16 stores/wave/round (128 KiB per block-round), R=%d rounds\n", R);
  for (int G : {1, 8, 32, 64, 128, 256}) {
    const float t0 = run<0, 0>(out, ld, G, R, 16, 0, sink), t1 = run<0, 1>(out, ld, G, R, 16, 0, sink);
    const double mb = (double)G * R * 131072 / 1e6;
    printf("G=%3d  plain %8.1f us  %7.1f GB/s/CU %7.2f TB/s | nt %8.1f us %7.1f GB/s/CU %7.2f TB/s\n", G, t0, mb / G / t0 * 1e3,
           mb / t0 / 1e3, t1, mb / G / t1 * 1e3, mb / t1 / 1e3);
  }
  printf("# part 2: G=256, per round S stores/wave + NM MFMAs/wave (NM=768 = one 256x256x768 tile)\n");
  for (int S : {16, 32})
    for (int NM : {768, 1536}) {
      const float ts = run<0, 1>(out, ld, 256, R, S, NM, sink), tm = run<1, 1>(out, ld, 256, R, S, NM, sink);
      const float tb = run<2, 1>(out, ld, 256, R, S, NM, sink), tser = run<3, 1>(out, ld, 256, R, S, NM, sink);
      printf("S=%2d NM=%4d  stores %8.1f  mfma %8.1f  both(no wait) %8.1f  both(vmcnt0) %8.1f us   [per round: %.2f %.2f %.2f %.2f]\n", S, NM,
             ts, tm, tb, tser, ts / R, tm / R, tb / R, tser / R);
    }
  return 0;
}
