// Probe (round 5, not part of liblc2is_hip.so): the K loop of an NT bf16 GEMM as TWO INDEPENDENT 4-wave blocks per CU —
// 128x256 tiles, 32-wide K slices in a 3-deep LDS-DMA ring (72 KiB per block), one barrier per slice, wave tile 128x64 (the
// shipped kernels' 128 accumulators) — against the shipped design's one 8-wave block per CU on 256x256 tiles with 64-wide stages.
// DESIGN §8.1 asks whether a second, independent block on each CU would fill the matrix pipe while the first sits in a
// barrier, a DMA wait or its epilogue.  K loop only: the epilogue stores the accumulators unstaged (checked against a host
// product on a small case), so the rate printed is an upper bound for a kernel of this shape.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I lc2is_amd/csrc -I include tools/probes/gemm_nt_h128/gemm_nt_h128.hip -o tools/probes/gemm_nt_h128/gemm_nt_h128.bin
// run:   tools/probes/gemm_nt_h128/gemm_nt_h128.bin
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

namespace {
constexpr int BM = 128, BN = 256, BK = 32, NST = 3;
constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE = A_BYTES + W_BYTES;   // 8 + 16 KiB
constexpr int TM = 8, TN = 4;

struct Args {
  const bf16_t* A; const bf16_t* W; float* C;
  int M, N, K, lda, ldw, ldc, ntiles, store;
};

// LDS image of a [rows][32] bf16 operand slice: two rows share a 128-byte line ("super row" sr = r >> 1), the eight 16-byte
// slots of a line are XOR-swizzled with sr — a fragment read (16 rows x 4 chunks of one wave instruction) then touches every
// bank group once, exactly as the shipped 128-byte-row images do.
__device__ __forceinline__ int img_off(int r, int c) { return (r >> 1) * 128 + ((((r & 1) << 2) | c) ^ ((r >> 1) & 7)) * 16; }

template <int NBLK>
__global__ __launch_bounds__(256, NBLK) void h128_kernel(Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, g = lane >> 4;
  const unsigned a_bytes = (unsigned)p.M * p.lda * 2u, w_bytes = (unsigned)p.N * p.ldw * 2u;
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, a_bytes), rsW = make_rsrc(p.W, w_bytes);
  const unsigned smem_a = (unsigned)(size_t)LDS_PTR(smem);
  const int ntn = p.N / BN;
  const int nk = p.K / BK;
  // this wave's 6 DMA pieces of a stage: pieces 0..7 are A (16 rows each), 8..23 W; wave w takes pieces w, w + 4, ...
  // lane l of a piece fills slot l & 7 of super row l >> 3 with the source chunk the swizzle maps there
  const int sl = lane & 7, srl = lane >> 3;
  const int cs = sl ^ srl;
  const int prow = 2 * srl + (cs >> 2), pch = cs & 3;
  // fragment bases (stage 0): X rows j * 16 + frow, W rows wid * 64 + i * 16 + frow, chunk g
  const unsigned x_base = smem_a + img_off(frow, g);
  const unsigned w_base = smem_a + A_BYTES + (wid * 64 / 2) * 128 + img_off(frow, g);

  for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
    const int m0 = (t / ntn) * BM, n0 = (t % ntn) * BN;
    int goff[6];
    unsigned ldst[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int pc = wid + 4 * q;
      if (pc < 8) goff[q] = ((m0 + pc * 16 + prow) * p.lda + pch * 8) * 2;
      else goff[q] = ((n0 + (pc - 8) * 16 + prow) * p.ldw + pch * 8) * 2;
      ldst[q] = (unsigned)(pc * 1024);   // A pieces then W pieces: the stage is one contiguous image
    }
    auto request = [&](int kt) __attribute__((always_inline)) {
      const unsigned so = smem_a + (unsigned)((kt % NST) * STAGE);
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int pc = wid + 4 * q;
        lds_dma16(pc < 8 ? rsA : rsW, so + ldst[q], goff[q], kt * BK * 2);
      }
    };
    f32x4_t acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    request(0);
    if (nk > 1) request(1);
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();          // slice kt has landed for every wave; every wave has left slice kt - 1
      if (kt + 2 < nk) request(kt + 2);
      const unsigned so = (unsigned)((kt % NST) * STAGE);
      bf16x8_t wf[TN], xf[TM];
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        bf16x8_t v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(w_base + so + i * 8 * 128));
        wf[i] = v;
      }
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        bf16x8_t v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(x_base + so + j * 8 * 128));
        xf[j] = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_barrier();            // (the next tile's first requests reuse slots 0 / 1)
    if (p.store) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const int m = m0 + j * 16 + frow, n = n0 + wid * 64 + i * 16 + g * 4;
          if (m < p.M) *(f32x4_t*)(p.C + (size_t)m * p.ldc + n) = acc[i][j];
        }
    } else {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) s += acc[i][j][0] + acc[i][j][3];
      if (s == 12345.678f) p.C[tid] = s;     // keeps the accumulators alive
    }
  }
}

float bf2f(bf16_t v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }
bf16_t f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); }
}  // namespace

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int NBLK>
int run(int M, int N, int K, bool check, int iters) {
  std::vector<bf16_t> ha((size_t)M * K), hw((size_t)N * K);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : ha) v = f2bf(rnd());
  for (auto& v : hw) v = f2bf(rnd() * 0.25f);
  bf16_t *dA, *dW; float* dC;
  CK(hipMalloc(&dA, ha.size() * 2 + 65536)); CK(hipMalloc(&dW, hw.size() * 2 + 65536)); CK(hipMalloc(&dC, (size_t)M * N * 4));
  CK(hipMemcpy(dA, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(dC, 0, (size_t)M * N * 4));
  Args a{dA, dW, dC, M, N, K, K, K, N, ((M + BM - 1) / BM) * (N / BN), check ? 1 : 0};
  const int lds = NST * STAGE;
  CK(hipFuncSetAttribute((const void*)h128_kernel<NBLK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int grid = 256 * NBLK < a.ntiles ? 256 * NBLK : a.ntiles;
  hipLaunchKernelGGL(h128_kernel<NBLK>, dim3(grid), dim3(256), lds, 0, a);
  CK(hipDeviceSynchronize());
  if (check) {
    std::vector<float> hc((size_t)M * N);
    CK(hipMemcpy(hc.data(), dC, hc.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int probe = 0; probe < 4000; ++probe) {
      s = s * 1664525u + 1013904223u; const int m = (s >> 4) % M;
      s = s * 1664525u + 1013904223u; const int n = (s >> 4) % N;
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)bf2f(ha[(size_t)m * K + k]) * bf2f(hw[(size_t)n * K + k]);
      worst = fmax(worst, fabs(ref - hc[(size_t)m * N + n]));
    }
    printf("check M=%d N=%d K=%d blocks/CU=%d: worst |err| over 4000 samples = %.3e\n", M, N, K, NBLK, worst);
  } else {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(h128_kernel<NBLK>, dim3(grid), dim3(256), lds, 0, a);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(h128_kernel<NBLK>, dim3(grid), dim3(256), lds, 0, a);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters;
    printf("M=%6d N=%5d K=%5d blocks/CU=%d tiles=%5d: %8.1f us  %7.1f TF/s\n", M, N, K, NBLK, a.ntiles, us, 2.0 * M * N * K / us * 1e-6);
  }
  hipFree(dA); hipFree(dW); hipFree(dC);
  return 0;
}

int main() {
  if (run<2>(300, 512, 160, true, 1)) return 1;
  if (run<2>(1024, 768, 3072, true, 1)) return 1;
  const int M = 32768;
  for (int rep = 0; rep < 2; ++rep) {
    if (run<2>(M, 3072, 768, false, 50)) return 1;    // fc1 / dfc2
    if (run<1>(M, 3072, 768, false, 50)) return 1;
    if (run<2>(M, 768, 3072, false, 50)) return 1;    // fc2 / dfc1
    if (run<2>(M, 2304, 768, false, 50)) return 1;    // qkv
    if (run<2>(8192, 8192, 8192, false, 10)) return 1;
    if (run<1>(8192, 8192, 8192, false, 10)) return 1;
  }
  return 0;
}
