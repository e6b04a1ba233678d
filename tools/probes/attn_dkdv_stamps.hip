// Probe (round 4): where does a 64-query tile of the attention dK/dV kernel spend its cycles?  A copy of the product kernel's
// common path (lc2is_amd/csrc/attention_bwd.hip: attn_bwd_dkdv_kernel<64, false>, no bias / causal / dropout: the ViT tower's
// instantiation — same LDS image, same LDS-DMA, same fragment reads, MFMAs and arithmetic) with s_memtime stamps between its
// segments; per wave the segment sums go to a debug buffer nobody else reads.  Also s_memrealtime around the loop (the clock).
//   segments per tile: [0] DMA request + stat loads of the next tile  [1] half 0: S / dP chains (8 b128 reads, 8 MFMAs)
//   [2] half 0: P / dS arithmetic (incl. the wait for the chains)  [3] half 0: dV^T / dK^T products (16 tr reads, 8 MFMAs)
//   [4][5][6] the same for half 1   [7] stat store + s_waitcnt vmcnt(0)   [8] s_barrier
// A stamp drains the LDS queue (s_waitcnt lgkmcnt(0) inside it), so read the SHARES, not the total.
// build: hipcc -O3 --offload-arch=gfx950 -I lc2is_amd/csrc -I include tools/probes/attn_dkdv_stamps.hip -o tools/probes/attn_dkdv_stamps.bin
#include "attn_common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstring>

namespace {
constexpr int D = 64, NKS = 4, NDT = 2;
using I = AttnCfg<D>;
constexpr int STAGE = 2 * I::TILE + 768;

__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
__device__ __forceinline__ bf16x8_t tr_frag3(const char* base, int addr_lo, int addr_hi) {
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)LDS_PTR(base + addr_lo));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)LDS_PTR(base + addr_hi));
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ bf16x8_t pack8(const f32x16_t& v, int base) {
  bf16x8_t r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)v[base + j];
  return r;
}

struct Args {
  const bf16_t *Q, *K, *V, *dO;
  bf16_t *dK, *dV;
  const float *lse2, *delta;
  int ld, B, H, S;
  float scale, scale_log2;
  unsigned long long* dbg;
};

__global__ __launch_bounds__(256, 2) void k(Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int hh = lane >> 5, l31 = lane & 31;
  const int nkb = (p.S + 127) / 128;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int bx = tile % nkb, head = (tile / nkb) % p.H, b = tile / (nkb * p.H);
  const int kcol = bx * 128 + wid * 32 + l31;
  const bool kok = kcol < p.S;
  const bool wave_active = (int)(bx * 128 + wid * 32) < p.S;
  const float INF = __builtin_inff();
  const int nqt = (p.S + 63) / 64;
  const unsigned bytes = (unsigned)p.B * p.S * p.ld * 2u;
  const __amdgpu_buffer_rsrc_t rsK = make_rsrc(p.K, bytes), rsV = make_rsrc(p.V, bytes);
  bf16x8_t kf[NKS], vf[NKS];
  {
    const int o = ((b * p.S + kcol) * p.ld + head * D + 8 * hh) * 2;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      kf[s] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rsK, kok ? o + s * 32 : -1, 0, 0));
      vf[s] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rsV, kok ? o + s * 32 : -1, 0, 0));
    }
  }
  int q_goff[I::PPW], g_goff[I::PPW];
#pragma unroll
  for (int j = 0; j < I::PPW; ++j) {
    const int row = I::RPP * (wid * I::PPW + j) + lane / I::SLOTS;
    const int ch = (lane % I::SLOTS) ^ I::swz(row);
    q_goff[j] = ch < I::CH ? ((b * p.S + row) * p.ld + head * D + ch * 8) * 2 : -1;
    g_goff[j] = q_goff[j];
  }
  float rstat = 0.f;
  auto gload = [&](int qt, char* stage) {
    attn_dma_tile<D>(p.Q, bytes, p.dO, bytes, stage, wid, q_goff, g_goff, qt * 64 * p.ld * 2, qt * 64 * p.ld * 2);
    if (tid < 128) {
      const int q = qt * 64 + (tid & 63);
      const size_t si = ((size_t)b * p.H + head) * p.S + q;
      const float* src = tid < 64 ? p.lse2 : p.delta;
      rstat = (q < p.S) ? src[si] : (tid < 64 ? INF : 0.f);
    }
  };
  auto lstore = [&](char* stage) {
    if (tid < 128) *(float*)(stage + 2 * I::TILE + tid * 4) = (tid < 64 && rstat == -INF) ? INF : rstat;
    wait_vm0();
  };
  const int q4 = (lane >> 2) & 3, p4 = lane & 3, cg = (lane >> 4) & 1;
  int row_addr[NKS], tr_lo[NDT], tr_hi[NDT];
#pragma unroll
  for (int s = 0; s < NKS; ++s) row_addr[s] = I::off(l31, 2 * s + hh);
#pragma unroll
  for (int d = 0; d < NDT; ++d) {
    const int e = 32 * d + 16 * cg + 4 * p4;
    tr_lo[d] = I::off(4 * hh + q4, e >> 3) + (e & 7) * 2;
    tr_hi[d] = I::off(4 * hh + q4 + 8, e >> 3) + (e & 7) * 2;
  }
  f32x16_t dkt[NDT], dvt[NDT];
#pragma unroll
  for (int d = 0; d < NDT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkt[d][r] = 0.f; dvt[d][r] = 0.f; }
  gload(0, smem);
  lstore(smem);
  __syncthreads();
  unsigned long long seg[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long c0 = stamp();
  for (int qt = 0; qt < nqt; ++qt) {
    unsigned long long t0 = stamp();
    const char* cur = smem + (qt & 1) * STAGE;
    char* nxt = smem + ((qt + 1) & 1) * STAGE;
    const bool more = (qt + 1) < nqt;
    if (more) gload(qt + 1, nxt);
    const float* lsev = (const float*)(cur + 2 * I::TILE);
    const float* delv = lsev + 64;
    unsigned long long t1 = stamp();
    seg[0] += t1 - t0;
    if (wave_active) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u == 1 && qt * 64 + 32 >= p.S) continue;
        t0 = stamp();
        f32x16_t sa, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
          const int a = row_addr[s] + 32 * u * I::PITCH;
          const bf16x8_t qfr = *(const bf16x8_t*)(cur + a);
          const bf16x8_t gfr = *(const bf16x8_t*)(cur + I::TILE + a);
          sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[s], sa, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gfr, vf[s], dp, 0, 0, 0);
        }
        t1 = stamp();
        seg[1 + 3 * u] += t1 - t0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4_t l4 = *(const f32x4_t*)(lsev + 32 * u + 8 * c + 4 * hh);
          const f32x4_t d4 = *(const f32x4_t*)(delv + 32 * u + 8 * c + 4 * hh);
          const f32x2_t sc2 = {p.scale_log2, p.scale_log2};
#pragma unroll
          for (int j = 0; j < 4; j += 2) {
            const f32x2_t s2 = f32x2_t{sa[4 * c + j], sa[4 * c + j + 1]} * sc2 - f32x2_t{l4[j], l4[j + 1]};
            const f32x2_t pr = {__builtin_amdgcn_exp2f(s2[0]), __builtin_amdgcn_exp2f(s2[1])};
            const f32x2_t ds = pr * (f32x2_t{dp[4 * c + j], dp[4 * c + j + 1]} - f32x2_t{d4[j], d4[j + 1]});
            sa[4 * c + j] = pr[0];
            sa[4 * c + j + 1] = pr[1];
            dp[4 * c + j] = ds[0];
            dp[4 * c + j + 1] = ds[1];
          }
        }
        bf16x8_t pf[2], dsf[2];
#pragma unroll
        for (int s2i = 0; s2i < 2; ++s2i) { pf[s2i] = pack8(sa, 8 * s2i); dsf[s2i] = pack8(dp, 8 * s2i); }
        asm volatile("" : "+v"(pf[0]), "+v"(pf[1]), "+v"(dsf[0]), "+v"(dsf[1]));   // the arithmetic is complete before the stamp
        t0 = stamp();
        seg[2 + 3 * u] += t0 - t1;
#pragma unroll
        for (int s2i = 0; s2i < 2; ++s2i) {
          const int roff = (32 * u + 16 * s2i) * I::PITCH;
#pragma unroll
          for (int d = 0; d < NDT; ++d) {
            const int lo = tr_lo[d] + roff, hi = tr_hi[d] + roff;
            const bf16x8_t gtf = tr_frag3(cur + I::TILE, lo, hi);
            const bf16x8_t qtf = tr_frag3(cur, lo, hi);
            dvt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gtf, pf[s2i], dvt[d], 0, 0, 0);
            dkt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf[s2i], dkt[d], 0, 0, 0);
          }
        }
        t1 = stamp();
        seg[3 + 3 * u] += t1 - t0;
      }
    }
    t0 = stamp();
    if (more) lstore(nxt);
    else wait_vm0();
    t1 = stamp();
    seg[7] += t1 - t0;
    __syncthreads();
    t0 = stamp();
    seg[8] += t0 - t1;
  }
  const unsigned long long c1 = stamp();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (kok) {
    bf16_t* krow = p.dK + (size_t)(b * p.S + kcol) * p.ld + head * D;
    bf16_t* vrow = p.dV + (size_t)(b * p.S + kcol) * p.ld + head * D;
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int col = 32 * d + 8 * c + 4 * hh;
        *reinterpret_cast<uint2*>(krow + col) = make_uint2(pack_bf16x2(dkt[d][4 * c] * p.scale, dkt[d][4 * c + 1] * p.scale),
                                                           pack_bf16x2(dkt[d][4 * c + 2] * p.scale, dkt[d][4 * c + 3] * p.scale));
        *reinterpret_cast<uint2*>(vrow + col) = make_uint2(pack_bf16x2(dvt[d][4 * c], dvt[d][4 * c + 1]),
                                                           pack_bf16x2(dvt[d][4 * c + 2], dvt[d][4 * c + 3]));
      }
  }
  if (lane == 0) {
    unsigned long long* o = p.dbg + ((size_t)blockIdx.x * 4 + wid) * 12;
#pragma unroll
    for (int i = 0; i < 9; ++i) o[i] = seg[i];
    o[9] = c1 - c0;
    o[10] = r1 - r0;
    o[11] = wave_active ? 1 : 0;
  }
}
}  // namespace

int main() {
  const int B = 32, H = 12, S = 1025, ld = 3 * H * D;
  const size_t n = (size_t)B * S * ld;
  std::vector<bf16_t> h(n);
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  for (auto& v : h) { float f = rnd() + rnd() + rnd(); unsigned u; memcpy(&u, &f, 4); v = (bf16_t)(u >> 16); }
  std::vector<float> lse((size_t)B * H * S), del((size_t)B * H * S);
  for (auto& v : lse) v = 12.f + rnd();        // log2-sum-exp of ~1025 scaled scores of unit-ish variance
  for (auto& v : del) v = 0.05f * rnd();
  bf16_t *qkv, *dO, *dqkv; float *dl, *dd; unsigned long long* dbg;
  const int nkb = (S + 127) / 128, grid = nkb * H * B;
  hipMalloc(&qkv, n * 2); hipMalloc(&dO, n * 2); hipMalloc(&dqkv, n * 2);
  hipMalloc(&dl, lse.size() * 4); hipMalloc(&dd, del.size() * 4); hipMalloc(&dbg, (size_t)grid * 4 * 12 * 8);
  hipMemcpy(qkv, h.data(), n * 2, hipMemcpyHostToDevice);
  hipMemcpy(dO, h.data(), n * 2, hipMemcpyHostToDevice);
  hipMemcpy(dl, lse.data(), lse.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dd, del.data(), del.size() * 4, hipMemcpyHostToDevice);
  Args a{qkv, qkv + H * D, qkv + 2 * H * D, dO, dqkv + H * D, dqkv + 2 * H * D, dl, dd, ld, B, H, S, 0.125f, 0.125f * 1.44269504f, dbg};
  const int LDS = 2 * STAGE;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), LDS, 0, a);   // settle the clock
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), LDS, 0, a);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> d((size_t)grid * 4 * 12);
  hipMemcpy(d.data(), dbg, d.size() * 8, hipMemcpyDeviceToHost);
  double seg[9] = {0}, cyc = 0, real = 0; long nw = 0;
  for (size_t w = 0; w < (size_t)grid * 4; ++w) {
    const unsigned long long* o = &d[w * 12];
    if (!o[11] || (w / 4) % nkb == (size_t)nkb - 1) continue;   // full key blocks only
    for (int i = 0; i < 9; ++i) seg[i] += (double)o[i];
    cyc += (double)o[9]; real += (double)o[10]; ++nw;
  }
  const int nqt = (S + 63) / 64;
  const char* names[9] = {"DMA request + stat loads", "half 0 chains (8 reads, 8 MFMA)", "half 0 arithmetic", "half 0 products (16 tr reads, 8 MFMA)",
                          "half 1 chains", "half 1 arithmetic", "half 1 products", "stat store + vmcnt(0)", "s_barrier"};
  printf("attn dK/dV, B*H = %d, S = %d, D = 64: kernel %.1f us (stamped build); loop %.0f cycles per wave = %.0f per 64-query tile; clock %.2f GHz\n",
         B * H, S, ms * 1e3 / 10, cyc / nw, cyc / nw / nqt, cyc / real * 0.1);
  double tot = 0;
  for (int i = 0; i < 9; ++i) tot += seg[i];
  for (int i = 0; i < 9; ++i) printf("  [%d] %-40s %7.0f cycles per tile  %5.1f %%\n", i, names[i], seg[i] / nw / nqt, 100.0 * seg[i] / tot);
  printf("  (matrix pipe: 32 MFMAs x 32 cycles = 1024 cycles per wave and tile; two waves share a SIMD)\n");
  return 0;
}
