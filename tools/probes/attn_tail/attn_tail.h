// The ragged last row of attention at S = 128 n + 1 (the ViT's 1025 = 1 CLS + 32 x 32 patch tokens), D = 64, no mask.
//
// With 128-query blocks (forward, dQ) and 128-key blocks (dK/dV) the 1025th token used to cost every (batch, head) a whole
// extra block in each of the three kernels — one of nine, sweeping all 17 tiles on the matrix cores for ONE row (11 % of the
// blocks for 0.1 % of the work; profiles/r02_*).  The kernels now launch 8 full blocks per (batch, head) plus one
// "tail block" that handles the last row with plain vector arithmetic: a thread owns the keys (queries) tid, tid + 256, ...,
// forms its dot products from 16-byte chunks of the rows (K, V, Q, dO stream through registers once; the single q / dO / k / v
// row of the block is broadcast from LDS), and the block reduces with DPP wave sums + one LDS step.  A tail block streams
// ~260 KiB that the (batch, head)'s main blocks have just pulled through L2 and runs ~20 x shorter than the block it replaces.
// Semantics = the same softmax(QK^T * scale) V row and its gradients (hf eager_attention_forward, modeling_clip.py:259-277);
// probabilities stay fp32 here (the tiled path rounds them to bf16 before P.V), which the parity tests' tolerances cover.
#pragma once
#include "common.h"

namespace {

constexpr int TAIL_D = 64;
constexpr int TAIL_MAX_PER_THREAD = 8;   // rows of the swept operand per thread: S <= 2048

__device__ __forceinline__ void tail_unpack8(const i32x4_t& v, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __builtin_bit_cast(float, ((unsigned)v[i]) << 16);
    f[2 * i + 1] = __builtin_bit_cast(float, ((unsigned)v[i]) & 0xffff0000u);
  }
}
// one 128-byte bf16 row -> 64 floats in LDS (the first 8 lanes of a wave carry it); caller synchronises
__device__ __forceinline__ void tail_row_to_lds(const bf16_t* row, float* dst, int tid) {
  if (tid < 8) {
    float f[8];
    tail_unpack8(*(const i32x4_t*)(row + 8 * tid), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[8 * tid + i] = f[i];
  }
}
__device__ __forceinline__ float tail_dot8(const i32x4_t& v, const float* bro) {   // packed chunk . broadcast row chunk (LDS)
  float f[8];
  tail_unpack8(v, f);
  const f32x4_t a = *(const f32x4_t*)bro, b = *(const f32x4_t*)(bro + 4);
  return f[0] * a[0] + f[1] * a[1] + f[2] * a[2] + f[3] * a[3] + f[4] * b[0] + f[5] * b[1] + f[6] * b[2] + f[7] * b[3];
}
// block-wide sum of `n` per-thread accumulators (n <= 128): wave sums by DPP, one LDS step; result valid in threads 0..n-1
template <int N>
__device__ __forceinline__ float tail_block_sum(float (&acc)[N], float* red /* [4][N] */, int tid) {
  const int lane = tid & 63, wid = tid >> 6;
  float mine = 0.f;
#pragma unroll
  for (int d = 0; d < N; ++d) {
    const float s = wave_sum(acc[d]);
    if ((d & 63) == lane && (d >> 6) == 0) mine = s;
    if (N > 64 && (d & 63) == lane && (d >> 6) == 1) red[wid * N + d] = s;
  }
  red[wid * N + lane] = mine;   // (N <= 64: lanes >= N write a zero nobody reads beyond N; N > 64: element `lane`)
  __syncthreads();
  float tot = 0.f;
  if (tid < N) tot = red[tid] + red[N + tid] + red[2 * N + tid] + red[3 * N + tid];
  return tot;
}

// ---- forward: O[row] and lse2[row] for row = Sq - 1 -----------------------------------------------------------------------
__device__ __forceinline__ void attn_fwd_tail_row(const bf16_t* Q, int ldq, const bf16_t* K, int ldk, const bf16_t* V, int ldv,
                                                  bf16_t* O, int ldo, float* lse2, int H, int Sq, int Sk, float scale_log2, int b,
                                                  int head, char* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float* qs = (float*)smem;           // [64]
  float* red = qs + 64;               // [4][64]
  float* stat = red + 256;            // [8]
  const int qr = Sq - 1;
  tail_row_to_lds(Q + (size_t)(b * Sq + qr) * ldq + head * TAIL_D, qs, tid);
  __syncthreads();
  const float NEG_INF = -__builtin_inff();
  float s[TAIL_MAX_PER_THREAD];
  float m = NEG_INF;
#pragma unroll
  for (int i = 0; i < TAIL_MAX_PER_THREAD; ++i) {
    const int k = tid + 256 * i;
    s[i] = NEG_INF;
    if (k < Sk) {
      const bf16_t* kr = K + (size_t)(b * Sk + k) * ldk + head * TAIL_D;
      float d = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) d += tail_dot8(*(const i32x4_t*)(kr + 8 * c), qs + 8 * c);
      s[i] = d * scale_log2;
    }
    m = fmaxf(m, s[i]);
  }
  m = wave_max(m);
  if (lane == 0) stat[wid] = m;
  __syncthreads();
  m = fmaxf(fmaxf(stat[0], stat[1]), fmaxf(stat[2], stat[3]));
  float acc[TAIL_D], lsum = 0.f;
#pragma unroll
  for (int d = 0; d < TAIL_D; ++d) acc[d] = 0.f;
#pragma unroll
  for (int i = 0; i < TAIL_MAX_PER_THREAD; ++i) {
    const int k = tid + 256 * i;
    if (k < Sk) {
      const float pr = __builtin_amdgcn_exp2f(s[i] - m);
      lsum += pr;
      const bf16_t* vr = V + (size_t)(b * Sk + k) * ldv + head * TAIL_D;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float f[8];
        tail_unpack8(*(const i32x4_t*)(vr + 8 * c), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[8 * c + e] += pr * f[e];
      }
    }
  }
  lsum = wave_sum(lsum);
  if (lane == 0) stat[4 + wid] = lsum;
  const float o = tail_block_sum<TAIL_D>(acc, red, tid);   // (its barrier also publishes stat[4..7])
  const float l = stat[4] + stat[5] + stat[6] + stat[7];
  if (tid < TAIL_D) O[(size_t)(b * Sq + qr) * ldo + head * TAIL_D + tid] = f32_to_bf16(o / l);
  if (tid == 0 && lse2) lse2[((size_t)b * H + head) * Sq + qr] = m + __builtin_amdgcn_logf(l);
}

// ---- backward, query side: dQ[row] and delta[row] for row = Sq - 1 -------------------------------------------------------
__device__ __forceinline__ void attn_bwd_tail_dq_row(const bf16_t* Q, int ldq, const bf16_t* K, int ldk, const bf16_t* V, int ldv,
                                                     const bf16_t* O, int ldo, const bf16_t* dO, int lddo, bf16_t* dQ, int lddq,
                                                     const float* lse2, float* delta, int H, int Sq, int Sk, float scale,
                                                     float scale_log2, int b, int head, char* smem) {
  const int tid = threadIdx.x;
  float* qs = (float*)smem;           // [64] q row
  float* gs = qs + 64;                // [64] dO row
  float* os = gs + 64;                // [64] O row
  float* red = os + 64;               // [4][64]
  const int qr = Sq - 1;
  const size_t tok = (size_t)(b * Sq + qr);
  tail_row_to_lds(Q + tok * ldq + head * TAIL_D, qs, tid);
  tail_row_to_lds(dO + tok * lddo + head * TAIL_D, gs, tid);
  tail_row_to_lds(O + tok * ldo + head * TAIL_D, os, tid);
  __syncthreads();
  float dl = 0.f;
#pragma unroll
  for (int d = 0; d < TAIL_D; ++d) dl += gs[d] * os[d];
  const size_t si = ((size_t)b * H + head) * Sq + qr;
  const float lse = lse2[si];
  if (tid == 0) delta[si] = dl;
  float acc[TAIL_D];
#pragma unroll
  for (int d = 0; d < TAIL_D; ++d) acc[d] = 0.f;
#pragma unroll 1
  for (int i = 0; i < TAIL_MAX_PER_THREAD; ++i) {
    const int k = tid + 256 * i;
    if (k < Sk) {
      const bf16_t* kr = K + (size_t)(b * Sk + k) * ldk + head * TAIL_D;
      const bf16_t* vr = V + (size_t)(b * Sk + k) * ldv + head * TAIL_D;
      i32x4_t kp[8];
      float sc = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        kp[c] = *(const i32x4_t*)(kr + 8 * c);
        sc += tail_dot8(kp[c], qs + 8 * c);
        dp += tail_dot8(*(const i32x4_t*)(vr + 8 * c), gs + 8 * c);
      }
      const float pr = __builtin_amdgcn_exp2f(sc * scale_log2 - lse);
      const float ds = pr * (dp - dl);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float f[8];
        tail_unpack8(kp[c], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[8 * c + e] += ds * f[e];
      }
    }
  }
  const float g = tail_block_sum<TAIL_D>(acc, red, tid);
  if (tid < TAIL_D) dQ[tok * lddq + head * TAIL_D + tid] = f32_to_bf16(g * scale);
}

// ---- backward, key side: dK[col], dV[col] for col = Sk - 1 (sums over every query; needs lse2 and delta of all rows) ---------
__device__ __forceinline__ void attn_bwd_tail_dkdv_col(const bf16_t* Q, int ldq, const bf16_t* K, int ldk, const bf16_t* V, int ldv,
                                                       const bf16_t* dO, int lddo, bf16_t* dK, int lddk, bf16_t* dV, int lddv,
                                                       const float* lse2, const float* delta, int H, int Sq, int Sk, float scale,
                                                       float scale_log2, int b, int head, char* smem) {
  const int tid = threadIdx.x;
  float* ks = (float*)smem;           // [64] k row
  float* vs = ks + 64;                // [64] v row
  float* red = vs + 64;               // [4][128]
  const int kc = Sk - 1;
  tail_row_to_lds(K + (size_t)(b * Sk + kc) * ldk + head * TAIL_D, ks, tid);
  tail_row_to_lds(V + (size_t)(b * Sk + kc) * ldv + head * TAIL_D, vs, tid);
  __syncthreads();
  // two passes over the queries, 32 columns of dV and dK each (128 fp32 accumulators per thread would spill beside the tiled path)
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    float acc[TAIL_D];   // dV[32 half .. +31] | dK[32 half .. +31]
#pragma unroll
    for (int d = 0; d < TAIL_D; ++d) acc[d] = 0.f;
#pragma unroll 1
    for (int i = 0; i < TAIL_MAX_PER_THREAD; ++i) {
      const int q = tid + 256 * i;
      if (q < Sq) {
        const bf16_t* qr = Q + (size_t)(b * Sq + q) * ldq + head * TAIL_D;
        const bf16_t* gr = dO + (size_t)(b * Sq + q) * lddo + head * TAIL_D;
        const size_t si = ((size_t)b * H + head) * Sq + q;
        i32x4_t qp[8], gp[8];
        float sc = 0.f, dp = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          qp[c] = *(const i32x4_t*)(qr + 8 * c);
          gp[c] = *(const i32x4_t*)(gr + 8 * c);
          sc += tail_dot8(qp[c], ks + 8 * c);
          dp += tail_dot8(gp[c], vs + 8 * c);
        }
        const float pr = __builtin_amdgcn_exp2f(sc * scale_log2 - lse2[si]);
        const float ds = pr * (dp - delta[si]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float f[8], h[8];
          tail_unpack8(half ? gp[4 + c] : gp[c], f);
          tail_unpack8(half ? qp[4 + c] : qp[c], h);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            acc[8 * c + e] += pr * f[e];
            acc[32 + 8 * c + e] += ds * h[e];
          }
        }
      }
    }
    if (half) __syncthreads();   // the first pass' readers are done with `red`
    const float g = tail_block_sum<TAIL_D>(acc, red, tid);
    if (tid < 32) dV[(size_t)(b * Sk + kc) * lddv + head * TAIL_D + 32 * half + tid] = f32_to_bf16(g);
    else if (tid < 64) dK[(size_t)(b * Sk + kc) * lddk + head * TAIL_D + 32 * half + (tid - 32)] = f32_to_bf16(g * scale);
  }
}

}  // namespace
