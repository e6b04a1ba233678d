// Probe: where does a K step of the 256x256 LDS-DMA NT GEMM main loop spend its cycles?  A copy of the product kernel's main loop
// (lc2is_amd/csrc/gemm_nt.hip: gemm_nt_dma_kernel<256,256,2,4>, same LDS image, same DMA, same fragment reads and MFMAs) with
// s_memtime stamps between its segments; per wave the segment sums go to a debug buffer nobody else reads, the accumulators to a
// dummy output.  Also stamps s_memrealtime (100 MHz) around the loop: the clock the chip holds.
//   segments per K step:  [0] DMA issue (8 pieces)   [1] K sub-step 0: 12 fragment reads + 32 MFMAs   [2] K sub-step 1
//                         [3] s_waitcnt vmcnt(0) (the next K tile's DMAs)   [4] s_barrier
// MODE 1: no DMA in the loop (tile 0 re-used)   MODE 2: DMA issued but never waited inside the loop (vmcnt only at the end; wrong data)
// build: hipcc -O3 --offload-arch=gfx950 -I lc2is_amd/csrc -I include tools/probes/gemm_step_stamps.hip -o tools/probes/gemm_step_stamps.bin
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {
constexpr int BM = 256, BN = 256, BK = 64, WAVES_N = 4, WM = 128, WN = 64, TM = 8, TN = 4, A_PIECES = 4, W_PIECES = 4;
constexpr int STAGE = (BM + BN) * 128;

__device__ __forceinline__ void dma_stage(const bf16_t* A, unsigned a_bytes, const bf16_t* W, unsigned w_bytes, char* buf, int wid,
                                          const int* a_goff, const int* w_goff, int kb) {
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A, a_bytes);
  const __amdgpu_buffer_rsrc_t rsW = make_rsrc(W, w_bytes);
#pragma unroll
  for (int j = 0; j < A_PIECES; ++j)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(buf + (wid * A_PIECES + j) * 1024), 16, a_goff[j] + kb, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < W_PIECES; ++j)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(buf + BM * 128 + (wid * W_PIECES + j) * 1024), 16, w_goff[j] + kb, 0, 0, 0);
}

__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

template <int MODE>
__global__ __launch_bounds__(512) void k(const bf16_t* A, int lda, const bf16_t* W, int ldw, float* out, int M, int N, int K,
                                         unsigned long long* dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int ntn = N / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
  const unsigned a_bytes = (unsigned)M * (unsigned)lda * 2u, w_bytes = (unsigned)N * (unsigned)ldw * 2u;
  const int lrow = lane >> 3, lch = (lane & 7) ^ (lane >> 3);
  int a_goff[A_PIECES], w_goff[W_PIECES];
#pragma unroll
  for (int j = 0; j < A_PIECES; ++j) a_goff[j] = ((m0 + 8 * (wid * A_PIECES + j) + lrow) * lda + lch * 8) * 2;
#pragma unroll
  for (int j = 0; j < W_PIECES; ++j) w_goff[j] = ((n0 + 8 * (wid * W_PIECES + j) + lrow) * ldw + lch * 8) * 2;
  f32x4_t acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, g = lane >> 4, sw = lane & 7;
  const int x_frag = (wm * WM + frow) * 128, w_frag = BM * 128 + (wn * WN + frow) * 128;
  const int kc_off0 = ((0 + g) ^ sw) << 4, kc_off1 = ((4 + g) ^ sw) << 4;
  const int nk = K / BK;
  dma_stage(A, a_bytes, W, w_bytes, smem, wid, a_goff, w_goff, 0);
  __syncthreads();
  unsigned long long seg[5] = {0, 0, 0, 0, 0};
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  unsigned long long t0 = stamp();
  const unsigned long long tstart = t0;
  for (int kt = 0; kt < nk; ++kt) {
    const char* cur = smem + ((MODE == 1 ? 0 : kt) & 1) * STAGE;
    if (MODE != 1 && kt + 1 < nk)
      dma_stage(A, a_bytes, W, w_bytes, smem + ((kt + 1) & 1) * STAGE, wid, a_goff, w_goff, (kt + 1) * BK * 2);
    unsigned long long t1 = stamp();
    seg[0] += t1 - t0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ko = ks ? kc_off1 : kc_off0;
      bf16x8_t xf[TM], wf[TN];
#pragma unroll
      for (int j = 0; j < TM; ++j) xf[j] = *(const bf16x8_t*)(cur + x_frag + j * 16 * 128 + ko);
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = *(const bf16x8_t*)(cur + w_frag + i * 16 * 128 + ko);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
      const unsigned long long t2 = stamp();
      seg[1 + ks] += t2 - t1;
      t1 = t2;
    }
    if (MODE != 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t3 = stamp();
    seg[3] += t3 - t1;
    __builtin_amdgcn_s_barrier();
    t0 = stamp();
    seg[4] += t0 - t3;
  }
  const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) {
    unsigned long long* d = dbg + ((size_t)blockIdx.x * 8 + wid) * 8;
#pragma unroll
    for (int i = 0; i < 5; ++i) d[i] = seg[i];
    d[5] = t0 - tstart;
    d[6] = rt1 - rt0;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[(size_t)blockIdx.x * 512 + tid] = s;
}

template <int MODE>
void run(const char* name, const bf16_t* A, const bf16_t* W, float* out, unsigned long long* dbg, int M, int N, int K) {
  auto kern = k<MODE>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  const int grid = (M / BM) * (N / BN);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 2 * STAGE, 0, A, K, W, K, out, M, N, K, dbg);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  std::vector<unsigned long long> h((size_t)grid * 64);
  hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
  double seg[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int b = 0; b < grid; ++b)
    for (int w = 0; w < 8; ++w)
      for (int i = 0; i < 7; ++i) seg[i] += (double)h[((size_t)b * 8 + w) * 8 + i];
  const double nw = (double)grid * 8, nk = K / BK;
  const double cyc = seg[5] / nw, ghz = seg[5] / (seg[6] * 10.0);   // memrealtime ticks are 10 ns
  printf("%-34s M=%d N=%d K=%d  kernel %.1f us (%.0f TF/s)  loop %.0f cyc = %.2f us at %.2f GHz; per K step: dma-issue %.0f | ks0 %.0f | ks1 %.0f | "
         "vmcnt %.0f | barrier %.0f  = %.0f cyc (%.2f us)\n",
         name, M, N, K, best * 1e3, 2.0 * M * N * K / best / 1e9, cyc, cyc / ghz / 1e3, ghz, seg[0] / nw / nk, seg[1] / nw / nk,
         seg[2] / nw / nk, seg[3] / nw / nk, seg[4] / nw / nk, cyc / nk, cyc / nk / ghz / 1e3);
}
}  // namespace

int main() {
  const int M = 32768;
  const size_t maxA = (size_t)M * 3072, maxW = (size_t)3072 * 3072;
  std::vector<bf16_t> ha(maxA), hw(maxW);
  srand(1);
  for (auto& v : ha) v = (bf16_t)(0x3c00 + (rand() & 0x83ff));   // random-ish bf16 in (-4, 4) with random signs / mantissas
  for (auto& v : hw) v = (bf16_t)(0x3a00 + (rand() & 0x81ff));
  bf16_t *A, *W; float* out; unsigned long long* dbg;
  hipMalloc(&A, maxA * 2); hipMalloc(&W, maxW * 2); hipMalloc(&out, (size_t)2048 * 512 * 4); hipMalloc(&dbg, (size_t)2048 * 64 * 8);
  hipMemcpy(A, ha.data(), maxA * 2, hipMemcpyHostToDevice);
  hipMemcpy(W, hw.data(), maxW * 2, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("full loop", A, W, out, dbg, M, 3072, 768);
    run<0>("full loop", A, W, out, dbg, M, 768, 3072);
    run<1>("no DMA in the loop", A, W, out, dbg, M, 3072, 768);
    run<1>("no DMA in the loop", A, W, out, dbg, M, 768, 3072);
    run<2>("DMA issued, never waited in loop", A, W, out, dbg, M, 3072, 768);
    run<2>("DMA issued, never waited in loop", A, W, out, dbg, M, 768, 3072);
  }
  return 0;
}
