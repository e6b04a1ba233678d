// Swin backbone kernels (model/encoder.py:121-131 -> hf:models/swin/modeling_swin.py).
//   * rows_gather: dst[r] = (map[r] >= 0 ? src[map[r]] : 0) (+ add[r]) over token rows.  With index maps built once
//     per (grid, window, shift) on the host side this one kernel is every re-layout of the path and its backward:
//     pad + cyclic shift + window partition (modeling_swin.py:546-550), window reverse + un-shift + un-pad + the
//     residual add (:558-567), and the 2x2 patch-merging concat (:318-321).  HBM-bound, 16-byte lanes.
//   * window attention (modeling_swin.py:373-398, 428-465): one wave per (window, head); S = ws*ws <= 64 tokens,
//     head_dim 32; logits get the per-head relative position bias [nH, S, S] and, for shifted blocks, the region mask
//     computed from the window's position (-100 where the 3x3 cyclic-shift regions of query and key differ, :584-607).
//     bf16 I/O, fp32 softmax.  Forward and backward run on the matrix cores (swin_attn_fwd_mfma_kernel: 16, and
//     swin_attn_bwd_mfma_kernel: 40 v_mfma_f32_32x32x16_bf16 per window-head, rows padded to 64 with zeros); the bias-table
//     gradient dS is accumulated per (head, window-chunk) in LDS in window order and written as partials that a second
//     launch sums in chunk order (bitwise reproducible, no atomics).  (The first, VALU forms — lane = query row, K / V
//     broadcast from LDS — measured 108 vs ~45 and 636 vs 98 us per launch and were removed in round 3.)
#include "common.h"
#include "lc2is_hip.h"
#include <cstdlib>

namespace {

constexpr int SW_D = 32;      // head_dim of every Swin variant (96/3, 128/4, 192/6 ...)
constexpr int SW_MAXS = 64;   // window tokens <= one wave

// LDS traffic of ONE wave is in order; this only stops the compiler from moving accesses across the hand-off point
#define SW_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// ---- generic row gather ------------------------------------------------------------------------------------------
template <bool SRC_BF16, bool DST_BF16>
__global__ __launch_bounds__(256) void rows_gather_kernel(const void* __restrict__ src_, int ld_src, void* dst_,
                                                           int ld_dst, const int* __restrict__ map,
                                                           const float* __restrict__ add, int ld_add, int rows,
                                                           int cols) {
  const int c4 = cols >> 2;  // 4-element groups per row
  const size_t total = (size_t)rows * c4;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int r = (int)(idx / c4), c = (int)(idx % c4) * 4;
    const int s = map[r];
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (s >= 0) {
      if (SRC_BF16) {
        const i32x2_t pk = *(const i32x2_t*)((const bf16_t*)src_ + (size_t)s * ld_src + c);
        v[0] = bf16_to_f32((bf16_t)(pk[0] & 0xffff)); v[1] = bf16_to_f32((bf16_t)((unsigned)pk[0] >> 16));
        v[2] = bf16_to_f32((bf16_t)(pk[1] & 0xffff)); v[3] = bf16_to_f32((bf16_t)((unsigned)pk[1] >> 16));
      } else {
        const f32x4_t f = *(const f32x4_t*)((const float*)src_ + (size_t)s * ld_src + c);
        v[0] = f[0]; v[1] = f[1]; v[2] = f[2]; v[3] = f[3];
      }
    }
    if (add) {
      const f32x4_t a = *(const f32x4_t*)(add + (size_t)r * ld_add + c);
      v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3];
    }
    if (DST_BF16) {
      i32x2_t pk = {(int)pack_bf16x2(v[0], v[1]), (int)pack_bf16x2(v[2], v[3])};
      *(i32x2_t*)((bf16_t*)dst_ + (size_t)r * ld_dst + c) = pk;
    } else {
      *(f32x4_t*)((float*)dst_ + (size_t)r * ld_dst + c) = f32x4_t{v[0], v[1], v[2], v[3]};
    }
  }
}

// ---- window attention ----------------------------------------------------------------------------------------------
struct SwinAttnArgs {
  const bf16_t* qkv; int ld;       // [nwin*S, 3C]: q | k | v, head h at columns h*32
  bf16_t* out; int ldo;            // forward: o [nwin*S, C]
  float* lse;                      // [nwin, nH, S]
  const float* bias;               // [nH, S, S]
  const bf16_t* dout; int lddo;    // backward: dO, O
  const bf16_t* o; int ld_o;
  bf16_t* dqkv; int lddq;          // [nwin*S, 3C]
  float* dbias_part;               // [nchunk, nH, S, S]
  int nwin, win_per_img, nwx, Hp, Wp, ws, shift, nH, C, chunk;
  float scale;
};

__device__ __forceinline__ int sw_region(int p, int extent, int ws, int shift) {
  return (p >= extent - ws) + (p >= extent - shift);
}

// LDS images are bf16 with a 144-byte row pitch (128 B of data + 16: conflict-free ds_read_b128 over 32 rows).
constexpr int SWM_P = 144;

__device__ __forceinline__ bf16x8_t swm_pack8(const f32x16_t& v, int base) {
  bf16x8_t r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)v[base + j];
  return r;
}

__device__ __forceinline__ float swm_dot8(const bf16x8_t& a, const bf16x8_t& b) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += (float)a[i] * (float)b[i];
  return s;
}

__global__ __launch_bounds__(64) void swin_attn_bwd_mfma_kernel(SwinAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x, l31 = lane & 31, hh = lane >> 5;
  const int S = a.ws * a.ws;
  const int h = blockIdx.x % a.nH, chunk = blockIdx.x / a.nH;
  char* QT = smem;                          // [32 d][SWM_P]  Q^T
  char* KT = QT + 32 * SWM_P;               //                K^T
  char* GT = KT + 32 * SWM_P;               //                dO^T
  char* DS = GT + 32 * SWM_P;               // [64 q][SWM_P]  dS
  float* Ls = (float*)(DS + 64 * SWM_P);    // [64] lse (+inf on the padding rows)
  float* Dp = Ls + 64;                      // [2][64] the two half-row partials of delta = dO . O
  int* Rid = (int*)(Dp + 128);              // [64] cyclic-shift region of each token of the window
  float* Acc = (float*)(Rid + 64);          // [S][S] bias-gradient accumulator of this block
  for (int idx = lane; idx < S * S; idx += 64) Acc[idx] = 0.f;
  const int w_begin = chunk * a.chunk;
  int w_end = w_begin + a.chunk;
  if (w_end > a.nwin) w_end = a.nwin;
  const float INF = __builtin_inff();
  const bf16x8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  const char* bias_h = (const char*)(a.bias + (size_t)h * S * S);
  for (int win = w_begin; win < w_end; ++win) {
    // (the per-lane offsets below do not depend on the window; computed outside the loop they would pin ~100 VGPRs, so an
    //  opaque zero ties them to the iteration)
    int opq = 0;
    asm volatile("" : "+v"(opq));
    const int hh_o = hh + opq, l31_o = l31 + opq;
    // ---- row fragments: tile t = rows 32t + l31, k step s = columns 16s + 8hh .. +7 of the head ----
    bf16x8_t qf[2][2], kf[2][2], vf[2][2], gf[2][2];
    float dpart[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int r = 32 * t + l31;
      const bool ok = r < S;
      const size_t row = (size_t)win * S + (ok ? r : 0);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int col = h * SW_D + 16 * s2 + 8 * hh;
        qf[t][s2] = ok ? *(const bf16x8_t*)(a.qkv + row * a.ld + col) : zero8;
        kf[t][s2] = ok ? *(const bf16x8_t*)(a.qkv + row * a.ld + a.C + col) : zero8;
        vf[t][s2] = ok ? *(const bf16x8_t*)(a.qkv + row * a.ld + 2 * a.C + col) : zero8;
        gf[t][s2] = ok ? *(const bf16x8_t*)(a.dout + row * a.lddo + col) : zero8;
        const bf16x8_t of = ok ? *(const bf16x8_t*)(a.o + row * a.ld_o + col) : zero8;
        dpart[t] += swm_dot8(gf[t][s2], of);
      }
    }
    const float lse_l = lane < S ? a.lse[((size_t)win * a.nH + h) * S + lane] : INF;
    const int widx = win % a.win_per_img, wy = widx / a.nwx, wx = widx % a.nwx;
    int rid_l = -1;
    if (lane < S && a.shift > 0)
      rid_l = sw_region(wy * a.ws + lane / a.ws, a.Hp, a.ws, a.shift) * 3 + sw_region(wx * a.ws + lane % a.ws, a.Wp, a.ws, a.shift);
    SW_LDS_SYNC();   // the previous window is done with the images
    Ls[lane] = lse_l;
    Rid[lane] = rid_l;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      Dp[hh * 64 + 32 * t + l31] = dpart[t];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int off = (16 * s2 + 8 * hh + e) * SWM_P + (32 * t + l31) * 2;
          *(__bf16*)(QT + off) = qf[t][s2][e];
          *(__bf16*)(KT + off) = kf[t][s2][e];
          *(__bf16*)(GT + off) = gf[t][s2][e];
        }
    }
    SW_LDS_SYNC();
    int rid_key[2];
    rid_key[0] = Rid[l31];
    rid_key[1] = Rid[32 + l31];
    f32x16_t dv[2], dk[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dv[t][r] = 0.f; dk[t][r] = 0.f; }
#pragma unroll
    for (int tq = 0; tq < 2; ++tq) {
      f32x16_t sa[2], dp[2];
#pragma unroll
      for (int tk = 0; tk < 2; ++tk) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[tk][r] = 0.f; dp[tk][r] = 0.f; }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          sa[tk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[tq][s2], kf[tk][s2], sa[tk], 0, 0, 0);
          dp[tk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gf[tq][s2], vf[tk][s2], dp[tk], 0, 0, 0);
        }
      }
      // ---- P, dS on the accumulator registers: register r = query row 32tq + 8(r>>2) + 4hh + (r&3), lane = key.
      // Branch-free: padding rows / columns read a clamped bias entry, get p = 0 and add their zero into a per-lane dummy
      // slot behind the accumulator (never two lanes on one LDS word). ----
      const int keyc0 = l31_o < S ? l31_o : S - 1, keyc1 = 32 + l31_o < S ? 32 + l31_o : S - 1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = 32 * tq + 8 * (r >> 2) + (r & 3) + 4 * hh_o;
        const bool q_ok = q < S;
        const int qc = q_ok ? q : S - 1;
        const float lse_q = Ls[q];
        const float del_q = Dp[q] + Dp[64 + q];
        const int rid_q = Rid[q];
#pragma unroll
        for (int tk = 0; tk < 2; ++tk) {
          const int key = 32 * tk + l31_o;
          const bool valid = q_ok && key < S;
          const float bz = *(const float*)(bias_h + (unsigned)((qc * S + (tk ? keyc1 : keyc0)) * 4));
          float sc = sa[tk][r] * a.scale + bz;
          sc += (a.shift > 0 && rid_q != rid_key[tk]) ? -100.0f : 0.f;
          const float p = valid ? __expf(sc - lse_q) : 0.f;   // (lse = +inf on padding rows: exp(-inf) = 0 as well)
          const float ds = p * (dp[tk][r] - del_q);
          float* slot = Acc + (valid ? q * S + key : S * S + lane);
          *slot += ds;
          sa[tk][r] = p;
          dp[tk][r] = ds;
          *(__bf16*)(DS + q * SWM_P + key * 2) = (__bf16)ds;
        }
      }
      // ---- dV += P^T.dO, dK += dS^T.Q over this tile of 32 queries ----
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int q0 = 32 * tq + 16 * s2 + 4 * hh;
        const s16x4_t g0 = *(const s16x4_t*)(GT + l31 * SWM_P + q0 * 2), g1 = *(const s16x4_t*)(GT + l31 * SWM_P + (q0 + 8) * 2);
        const s16x4_t x0 = *(const s16x4_t*)(QT + l31 * SWM_P + q0 * 2), x1 = *(const s16x4_t*)(QT + l31 * SWM_P + (q0 + 8) * 2);
        const s16x8_t gv = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
        const s16x8_t xv = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        const bf16x8_t gB = __builtin_bit_cast(bf16x8_t, gv), qB = __builtin_bit_cast(bf16x8_t, xv);
#pragma unroll
        for (int tk = 0; tk < 2; ++tk) {
          dv[tk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(swm_pack8(sa[tk], 8 * s2), gB, dv[tk], 0, 0, 0);
          dk[tk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(swm_pack8(dp[tk], 8 * s2), qB, dk[tk], 0, 0, 0);
        }
      }
    }
    SW_LDS_SYNC();   // the dS image is complete
    // ---- dQ = dS.K (rows q from the dS image, K^T image), then the three stores: lane = column d, registers = rows ----
#pragma unroll
    for (int tq = 0; tq < 2; ++tq) {
      f32x16_t dq;
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[r] = 0.f;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const bf16x8_t A = *(const bf16x8_t*)(DS + (32 * tq + l31) * SWM_P + (16 * s4 + 8 * hh) * 2);
        const bf16x8_t B = *(const bf16x8_t*)(KT + l31 * SWM_P + (16 * s4 + 8 * hh) * 2);
        dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, dq, 0, 0, 0);
      }
      char* out_w = (char*)(a.dqkv + (size_t)win * S * a.lddq + h * SW_D);   // uniform base of this (window, head)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = 32 * tq + 8 * (r >> 2) + (r & 3) + 4 * hh_o;
        if (q < S) {
          bf16_t* dst = (bf16_t*)(out_w + (unsigned)((q * a.lddq + l31_o) * 2));
          dst[0] = f32_to_bf16(dq[r] * a.scale);
          dst[a.C] = f32_to_bf16(dk[tq][r] * a.scale);   // (tile index = key tile here: same row numbering)
          dst[2 * a.C] = f32_to_bf16(dv[tq][r]);
        }
      }
    }
  }
  SW_LDS_SYNC();
  if (a.dbias_part) {
    float* dst = a.dbias_part + ((size_t)chunk * a.nH + h) * S * S;
    for (int idx = lane; idx < S * S; idx += 64) dst[idx] = Acc[idx];
  }
}

// ---- forward on the matrix cores: "query on the lane" (the orientation of attention_fwd.hip) --------------------------------
// One wave per block, block (h, chunk) walks its windows; the head's [S][S] bias tile is loaded into LDS once per block.
//   S^T[key][q] = K.Q^T  (A = K row fragments, B = Q row fragments, both straight from global memory): lane = query, the
//   registers of a tile = 16 keys, so the softmax statistics are per-lane loops plus one lane^32 exchange;
//   O^T[d][q] += V^T.P^T with the exp'd score registers, packed to bf16, as the B operand and V^T read from an LDS image in
//   the matching key order (keys key0+{0..3}, key0+8+{0..3}, key0 = 32tk + 16s2 + 4hh).  16 MFMAs per (window, head).
__global__ __launch_bounds__(64) void swin_attn_fwd_mfma_kernel(SwinAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x, l31 = lane & 31, hh = lane >> 5;
  const int S = a.ws * a.ws;
  const int h = blockIdx.x % a.nH, chunk = blockIdx.x / a.nH;
  char* VT = smem;                              // [32 d][SWM_P]  V^T (bf16)
  int* Rid = (int*)(VT + 32 * SWM_P);           // [64]
  float* Bs = (float*)(Rid + 64);               // [S][S] bias of head h
  for (int idx = lane; idx < S * S; idx += 64) Bs[idx] = a.bias[(size_t)h * S * S + idx];
  const int w_begin = chunk * a.chunk;
  int w_end = w_begin + a.chunk;
  if (w_end > a.nwin) w_end = a.nwin;
  const float NEG = -__builtin_inff();
  const bf16x8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int win = w_begin; win < w_end; ++win) {
    int opq = 0;   // ties the per-lane offsets to the iteration (see the backward kernel)
    asm volatile("" : "+v"(opq));
    const int hh_o = hh + opq, l31_o = l31 + opq;
    bf16x8_t qf[2][2], kf[2][2], vf[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int r = 32 * t + l31_o;
      const bool ok = r < S;
      const size_t row = (size_t)win * S + (ok ? r : 0);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int col = h * SW_D + 16 * s2 + 8 * hh_o;
        qf[t][s2] = ok ? *(const bf16x8_t*)(a.qkv + row * a.ld + col) : zero8;
        kf[t][s2] = ok ? *(const bf16x8_t*)(a.qkv + row * a.ld + a.C + col) : zero8;
        vf[t][s2] = ok ? *(const bf16x8_t*)(a.qkv + row * a.ld + 2 * a.C + col) : zero8;
      }
    }
    const int widx = win % a.win_per_img, wy = widx / a.nwx, wx = widx % a.nwx;
    int rid_l = -1;
    if (lane < S && a.shift > 0)
      rid_l = sw_region(wy * a.ws + lane / a.ws, a.Hp, a.ws, a.shift) * 3 + sw_region(wx * a.ws + lane % a.ws, a.Wp, a.ws, a.shift);
    SW_LDS_SYNC();   // the previous window is done with the image
    Rid[lane] = rid_l;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          *(__bf16*)(VT + (16 * s2 + 8 * hh + e) * SWM_P + (32 * t + l31) * 2) = vf[t][s2][e];
    SW_LDS_SYNC();
#pragma unroll
    for (int tq = 0; tq < 2; ++tq) {
      const int q = 32 * tq + l31_o;
      const bool q_ok = q < S;
      const int qc = q_ok ? q : S - 1;
      const int rid_q = Rid[q];
      f32x16_t st[2];
#pragma unroll
      for (int tk = 0; tk < 2; ++tk) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[tk][r] = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) st[tk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[tk][s2], qf[tq][s2], st[tk], 0, 0, 0);
      }
      float m = NEG;
#pragma unroll
      for (int tk = 0; tk < 2; ++tk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = 32 * tk + 8 * (r >> 2) + (r & 3) + 4 * hh_o;
          const int kc = key < S ? key : S - 1;
          float sc = st[tk][r] * a.scale + Bs[qc * S + kc];
          sc += (a.shift > 0 && Rid[key] != rid_q) ? -100.0f : 0.f;
          sc = key < S ? sc : NEG;
          st[tk][r] = sc;
          m = fmaxf(m, sc);
        }
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      float l = 0.f;
#pragma unroll
      for (int tk = 0; tk < 2; ++tk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __expf(st[tk][r] - m);
          st[tk][r] = p;
          l += p;
        }
      l += __shfl_xor(l, 32, 64);
      f32x16_t ot;
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[r] = 0.f;
#pragma unroll
      for (int tk = 0; tk < 2; ++tk)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int key0 = 32 * tk + 16 * s2 + 4 * hh;
          const s16x4_t v0 = *(const s16x4_t*)(VT + l31 * SWM_P + key0 * 2), v1 = *(const s16x4_t*)(VT + l31 * SWM_P + (key0 + 8) * 2);
          const s16x8_t vv = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          ot = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vv), swm_pack8(st[tk], 8 * s2), ot, 0, 0, 0);
        }
      if (q_ok) {
        const float inv = 1.0f / l;
        bf16_t* orow = a.out + ((size_t)win * S + q) * a.ldo + h * SW_D;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          i32x2_t pk = {(int)pack_bf16x2(ot[4 * c] * inv, ot[4 * c + 1] * inv), (int)pack_bf16x2(ot[4 * c + 2] * inv, ot[4 * c + 3] * inv)};
          *(i32x2_t*)(orow + 8 * c + 4 * hh) = pk;
        }
        if (a.lse && hh == 0) a.lse[((size_t)win * a.nH + h) * S + q] = m + __logf(l);
      }
    }
  }
}

// Relative-position-bias TABLE gradient: dtable[t][h] = sum over the (query, key) pairs whose relative offset is t of
// dbias[h][pair] (modeling_swin.py:371-383 gathers table[index] in the forward).  The pairs of every offset are listed in a
// CSR index built once on the host (offsets[T+1], positions sorted by offset, ascending pair id inside an offset): one
// thread per (t, h) adds its <= ws^2 terms in that fixed order, so the result is reproducible.
__global__ __launch_bounds__(256) void swin_table_grad_kernel(const float* __restrict__ dbias, const int* __restrict__ offs,
                                                               const int* __restrict__ pos, float* dtable, int nH, int SS,
                                                               int T, int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= T * nH) return;
  const int t = i / nH, h = i % nH;
  float s = 0.f;
  for (int e = offs[t]; e < offs[t + 1]; ++e) s += dbias[(size_t)h * SS + pos[e]];
  dtable[i] = accumulate ? dtable[i] + s : s;
}

// sum of the per-chunk partials in chunk order: 32 columns x 8 chunk groups per block, groups combined in a fixed order
__global__ __launch_bounds__(256) void swin_dbias_reduce_kernel(const float* __restrict__ part, int nchunk, size_t n,
                                                                 float* out, int accumulate) {
  __shared__ float red[8][32];
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const size_t i = (size_t)blockIdx.x * 32 + col;
  float s = 0.f;
  if (i < n)
    for (int c = grp; c < nchunk; c += 8) s += part[(size_t)c * n + i];
  red[grp][col] = s;
  __syncthreads();
  if (grp == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) t += red[g][col];
    out[i] = accumulate ? out[i] + t : t;
  }
}

inline int sw_chunks(int nwin, int nH) {
  int nchunk = 1792 / (nH > 0 ? nH : 1);   // ~7 single-wave blocks per CU (LDS-limited residency)
  if (nchunk < 1) nchunk = 1;
  if (nchunk > nwin) nchunk = nwin;
  return nchunk;
}

inline int sw_check(int nwin, int win_per_img, int nwx, int Hp, int Wp, int ws, int shift, int nH, int C) {
  if (nwin <= 0 || win_per_img <= 0 || nwx <= 0 || ws <= 0 || nH <= 0) return LC2IS_ERR_SHAPE;
  if (ws * ws > SW_MAXS || C != nH * SW_D) return LC2IS_ERR_UNSUPPORTED;
  if (Hp % ws || Wp % ws || nwx != Wp / ws || win_per_img != (Hp / ws) * nwx || nwin % win_per_img) return LC2IS_ERR_SHAPE;
  if (shift < 0 || shift >= ws) return LC2IS_ERR_SHAPE;
  return LC2IS_OK;
}

}  // namespace

extern "C" int lc2is_rows_gather(const void* src, int ld_src, int src_bf16, void* dst, int ld_dst, int dst_bf16,
                                 const int* map, const float* add, int ld_add, int rows, int cols,
                                 lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!src || !dst || !map) return LC2IS_ERR_NULL;
  if (rows <= 0 || cols <= 0 || cols % 4 || ld_src < cols || ld_dst < cols || ld_src % 4 || ld_dst % 4) return LC2IS_ERR_SHAPE;
  if (add && (ld_add < cols || ld_add % 4)) return LC2IS_ERR_SHAPE;
  size_t g = ((size_t)rows * (cols / 4) + 255) / 256;
  if (g > 16384) g = 16384;
#define RG(SB, DB) hipLaunchKernelGGL((rows_gather_kernel<SB, DB>), dim3((int)g), dim3(256), 0, stream, src, ld_src, dst, \
                                      ld_dst, map, add, ld_add, rows, cols)
  if (src_bf16 && dst_bf16) RG(true, true);
  else if (src_bf16) RG(true, false);
  else if (dst_bf16) RG(false, true);
  else RG(false, false);
#undef RG
  return lc2is_check_launch();
}

extern "C" int lc2is_swin_attn_fwd(const void* qkv, int ld, void* out, int ldo, float* lse, const float* bias, int nwin,
                                   int win_per_img, int nwx, int Hp, int Wp, int ws, int shift, int nH, int C,
                                   float scale, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!qkv || !out || !bias) return LC2IS_ERR_NULL;
  int rc = sw_check(nwin, win_per_img, nwx, Hp, Wp, ws, shift, nH, C);
  if (rc) return rc;
  if (ld < 3 * C || ldo < C || ld % 8 || ldo % 8) return LC2IS_ERR_SHAPE;
  SwinAttnArgs a{};
  a.qkv = (const bf16_t*)qkv; a.ld = ld; a.out = (bf16_t*)out; a.ldo = ldo; a.lse = lse; a.bias = bias;
  a.nwin = nwin; a.win_per_img = win_per_img; a.nwx = nwx; a.Hp = Hp; a.Wp = Wp; a.ws = ws; a.shift = shift; a.nH = nH;
  a.C = C; a.scale = scale;
  // matrix-core form: single-wave blocks, ~14 KB of LDS each (8 resident per CU), one round of them
  int nchunk = 2048 / nH > 0 ? 2048 / nH : 1;
  if (nchunk > nwin) nchunk = nwin;
  a.chunk = (nwin + nchunk - 1) / nchunk;
  const int nchunk_eff = (nwin + a.chunk - 1) / a.chunk;
  const int S_ = ws * ws;
  const int lds_mfma = 32 * SWM_P + 64 * 4 + S_ * S_ * (int)sizeof(float);
  hipLaunchKernelGGL(swin_attn_fwd_mfma_kernel, dim3(nchunk_eff * nH), dim3(64), lds_mfma, stream, a);
  return lc2is_check_launch();
}

extern "C" size_t lc2is_swin_attn_bwd_workspace_bytes(int nwin, int ws, int nH) {
  if (nwin <= 0 || ws <= 0 || nH <= 0) return 0;
  return (size_t)sw_chunks(nwin, nH) * nH * ws * ws * ws * ws * sizeof(float);
}

extern "C" int lc2is_swin_attn_bwd(const void* qkv, int ld, const void* o, int ld_o, const void* dout, int lddo,
                                   const float* lse, const float* bias, void* dqkv, int lddq, float* dbias,
                                   int accumulate_dbias, int nwin, int win_per_img, int nwx, int Hp, int Wp, int ws,
                                   int shift, int nH, int C, float scale, void* workspace, size_t workspace_bytes,
                                   lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!qkv || !o || !dout || !lse || !bias || !dqkv) return LC2IS_ERR_NULL;
  int rc = sw_check(nwin, win_per_img, nwx, Hp, Wp, ws, shift, nH, C);
  if (rc) return rc;
  if (ld < 3 * C || lddq < 3 * C || ld_o < C || lddo < C || ld % 8 || lddq % 8 || ld_o % 8 || lddo % 8) return LC2IS_ERR_SHAPE;
  if (dbias && (!workspace || workspace_bytes < lc2is_swin_attn_bwd_workspace_bytes(nwin, ws, nH))) return LC2IS_ERR_WORKSPACE;
  int nchunk = sw_chunks(nwin, nH);
  {   // the kernel holds 34 KB of LDS and ~300 registers: four single-wave blocks per CU, one round of them
    const int c4 = 1024 / nH > 0 ? 1024 / nH : 1;
    if (nchunk > c4) nchunk = c4;
  }
  SwinAttnArgs a{};
  a.qkv = (const bf16_t*)qkv; a.ld = ld; a.o = (const bf16_t*)o; a.ld_o = ld_o; a.dout = (const bf16_t*)dout; a.lddo = lddo;
  a.lse = const_cast<float*>(lse); a.bias = bias; a.dqkv = (bf16_t*)dqkv; a.lddq = lddq;
  a.dbias_part = dbias ? (float*)workspace : nullptr;
  a.nwin = nwin; a.win_per_img = win_per_img; a.nwx = nwx; a.Hp = Hp; a.Wp = Wp; a.ws = ws; a.shift = shift; a.nH = nH;
  a.C = C; a.scale = scale; a.chunk = (nwin + nchunk - 1) / nchunk;
  const int S_ = ws * ws;
  const int lds_mfma = (3 * 32 + 64) * SWM_P + (64 + 128 + 64) * 4 + (S_ * S_ + 64) * (int)sizeof(float);
  static DevOnce attr_set;
  if (attr_set.need()) {
    if (hipFuncSetAttribute((const void*)swin_attn_bwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess)
      return LC2IS_ERR_LAUNCH;
    attr_set.done();
  }
  const int nchunk_eff = (nwin + a.chunk - 1) / a.chunk;
  hipLaunchKernelGGL(swin_attn_bwd_mfma_kernel, dim3(nchunk_eff * nH), dim3(64), lds_mfma, stream, a);
  rc = lc2is_check_launch();
  if (rc || !dbias) return rc;
  const size_t n = (size_t)nH * ws * ws * ws * ws;
  hipLaunchKernelGGL(swin_dbias_reduce_kernel, dim3((int)((n + 31) / 32)), dim3(256), 0, stream,
                     (const float*)workspace, nchunk_eff, n, dbias, accumulate_dbias);
  return lc2is_check_launch();
}

extern "C" int lc2is_swin_bias_table_grad(const float* dbias, const int* offsets, const int* positions, float* dtable, int nH,
                                          int SS, int T, int accumulate, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!dbias || !offsets || !positions || !dtable) return LC2IS_ERR_NULL;
  if (nH <= 0 || SS <= 0 || T <= 0) return LC2IS_ERR_SHAPE;
  hipLaunchKernelGGL(swin_table_grad_kernel, dim3((T * nH + 255) / 256), dim3(256), 0, stream, dbias, offsets, positions,
                     dtable, nH, SS, T, accumulate);
  return lc2is_check_launch();
}
