// Segmentation head tail: xS upsample of the per-class score map fused with softmax cross-entropy,
// forward AND backward in one pass (gfx950).
// replaces: F.interpolate(mode="bicubic", scale_factor=4) + rearrange + matmul + nn.CrossEntropyLoss
//   (reference model/model.py:41-53 + evaluate.py:68 / engine.py:94) and AuxiliaryLoss.forward
//   (model/loss.py:17-21: bilinear resize, then CE), plus their autograd.
//
// The upsample is linear with taps summing to 1, so it commutes with TextToPatch.visual and the prototype
// matmul: the host computes class scores at LOW resolution [B,h,w,Cp] (channels-last fp32,
// Cp = padded class count) with two small MFMA GEMMs and this kernel interpolates 151 channels instead of
// 768 — the [B,16384,768] tensor (25 MB/img) is never materialised.
//
// Work split: a block owns a 16x16 tile of OUTPUT pixels.  The <=7x7 low-res footprint of the tile
// (S >= 4) is staged in LDS once; each wave then walks 64 output pixels with the 64 lanes spread over
// CHANNELS (3 per lane, C <= 192): LDS reads, the softmax reductions and the gradient scatter are all
// lane-contiguous (conflict-free ds_read / ds_add_f32).  The gradient wrt the low-res scores is accumulated
// in an LDS mirror of the footprint and flushed with one fp32 atomic add per footprint element — 256-byte
// contiguous segments, the shape global float atomics run at full rate.  HBM traffic: the low-res scores
// and labels once, the low-res gradient once.
#include "common.h"
#include "lc2is_hip.h"

namespace {

constexpr int HT = 16;        // output tile edge
constexpr int FMAX = 8;       // max footprint edge for S >= 4 (16/S + 4 bicubic rows)
constexpr int HEAD_THREADS = 512;
constexpr int CMAX = 192;

__device__ __forceinline__ float cubic1(float x) { return ((1.25f * x - 2.25f) * x) * x + 1.f; }          // A=-0.75
__device__ __forceinline__ float cubic2(float x) { return ((-0.75f * x + 3.75f) * x - 6.f) * x + 3.f; }

// taps of one output coordinate: up to 4 (index, weight) pairs, indices clamped to [0, n-1]
struct Taps { int idx[4]; float w[4]; };

__device__ __forceinline__ Taps make_taps(int dst, float inv_scale, int n_in, int mode) {
  Taps t;
  if (mode == LC2IS_INTERP_BICUBIC) {
    const float src = inv_scale * ((float)dst + 0.5f) - 0.5f;
    const float fl = floorf(src);
    const float tt = src - fl;
    const int i0 = (int)fl;
    t.w[0] = cubic2(tt + 1.f); t.w[1] = cubic1(tt); t.w[2] = cubic1(1.f - tt); t.w[3] = cubic2(2.f - tt);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int ii = i0 - 1 + k;
      t.idx[k] = ii < 0 ? 0 : (ii > n_in - 1 ? n_in - 1 : ii);
    }
  } else {  // bilinear, align_corners=False
    float src = inv_scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    const int i0 = (int)src;
    const int i1 = i0 < n_in - 1 ? i0 + 1 : i0;
    const float l1 = src - (float)i0;
    t.idx[0] = i0; t.w[0] = 1.f - l1; t.idx[1] = i1; t.w[1] = l1;
    t.idx[2] = i0; t.w[2] = 0.f; t.idx[3] = i0; t.w[3] = 0.f;
  }
  return t;
}

struct HeadArgs {
  const float* lo; int ld;       // [B, h, w, ld] scores, C valid channels
  const int64_t* labels;         // [B, H, W]
  float* dlo;                    // [B, h, w, ld] gradient accumulator (pre-zeroed) or null
  float* hi_out;                 // [B, C, H, W] upsampled scores (eval) or null
  float* loss_sum;               // [2]: sum of per-pixel losses, number of counted pixels (atomic)
  int B, h, w, H, W, C, S, mode;
  long ignore_index;
  float gscale;                  // dlo = gscale * (softmax - onehot)
};

__device__ __forceinline__ void lds_add(float* p, float v) {
  __hip_atomic_fetch_add((__attribute__((address_space(3))) float*)LDS_PTR(p), v, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int MODE>
__global__ __launch_bounds__(HEAD_THREADS) void head_ce_kernel(HeadArgs p) {
  constexpr int NTAP = (MODE == LC2IS_INTERP_BICUBIC) ? 4 : 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int tiles_x = (p.W + HT - 1) / HT, tiles_y = (p.H + HT - 1) / HT;
  const int b = blockIdx.x / (tiles_x * tiles_y);
  const int ty = (blockIdx.x / tiles_x) % tiles_y, tx = blockIdx.x % tiles_x;
  const int Y0 = ty * HT, X0 = tx * HT;
  const float inv_scale = 1.f / (float)p.S;

  // footprint origin: lowest tap index of the tile's first row/col, highest of its last
  const Taps ty0 = make_taps(Y0, inv_scale, p.h, MODE);
  const Taps ty1 = make_taps(min(Y0 + HT - 1, p.H - 1), inv_scale, p.h, MODE);
  const Taps tx0 = make_taps(X0, inv_scale, p.w, MODE);
  const Taps tx1 = make_taps(min(X0 + HT - 1, p.W - 1), inv_scale, p.w, MODE);
  const int fy0 = ty0.idx[0], fx0 = tx0.idx[0];
  int fy1 = ty1.idx[0], fx1 = tx1.idx[0];
#pragma unroll
  for (int k = 1; k < 4; ++k) { fy1 = max(fy1, ty1.idx[k]); fx1 = max(fx1, tx1.idx[k]); }
  const int FH = fy1 - fy0 + 1, FW = fx1 - fx0 + 1;  // <= FMAX by construction (checked on the host)

  const int Cp = p.ld;  // channel pitch in LDS == global pitch (multiple of 64 floats keeps rows bank-aligned)
  float* s_lo = (float*)smem;
  float* s_dlo = s_lo + FMAX * FMAX * Cp;
  const int fsize = FH * FW * Cp;
  for (int i = tid * 4; i < fsize; i += HEAD_THREADS * 4) {
    const int cell = i / Cp, c = i % Cp;
    const int fy = cell / FW, fx = cell % FW;
    const float4 v = *reinterpret_cast<const float4*>(
        p.lo + (((size_t)b * p.h + fy0 + fy) * p.w + fx0 + fx) * p.ld + c);
    *reinterpret_cast<float4*>(s_lo + i) = v;
    if (p.dlo) *reinterpret_cast<float4*>(s_dlo + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();

  // this wave's 32 pixels: rows 2*wid, 2*wid+1 of the tile, 16 columns each; lane <-> pixel for labels
  const int py_l = 2 * wid + ((lane >> 4) & 1), px_l = lane & 15;
  const int Yl = Y0 + py_l, Xl = X0 + px_l;
  int my_label = -1;
  const bool my_valid = (Yl < p.H && Xl < p.W);
  if (my_valid && p.labels) {
    const int64_t lab64 = p.labels[((size_t)b * p.H + Yl) * p.W + Xl];
    my_label = (lab64 == (int64_t)p.ignore_index || lab64 < 0 || lab64 >= p.C) ? -1 : (int)lab64;
  }

  const bool c_ok[3] = {lane < p.C, lane + 64 < p.C, lane + 128 < p.C};
  float loss_acc = 0.f, cnt_acc = 0.f;

  for (int px = 0; px < 32; ++px) {
    const int Y = Y0 + 2 * wid + (px >> 4), X = X0 + (px & 15);
    if (Y >= p.H || X >= p.W) continue;  // wave-uniform
    const int label = __builtin_amdgcn_readlane(my_label, px);   // wave-uniform lane index: v_readlane, not an LDS bpermute
    const Taps ay = make_taps(Y, inv_scale, p.h, MODE);
    const Taps ax = make_taps(X, inv_scale, p.w, MODE);
    float v[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NTAP; ++i) {
      const float* rowp = s_lo + ((ay.idx[i] - fy0) * FW - fx0) * Cp + lane;
#pragma unroll
      for (int j = 0; j < NTAP; ++j) {
        const float wgt = ay.w[i] * ax.w[j];
        const float* cp = rowp + ax.idx[j] * Cp;
        v[0] += wgt * cp[0];
        v[1] += wgt * cp[64];
        v[2] += wgt * cp[128];
      }
    }
    if (p.hi_out) {
      const size_t plane = (size_t)p.H * p.W;
      float* o = p.hi_out + ((size_t)b * p.C) * plane + (size_t)Y * p.W + X;
      if (c_ok[0]) o[(size_t)lane * plane] = v[0];
      if (c_ok[1]) o[(size_t)(lane + 64) * plane] = v[1];
      if (c_ok[2]) o[(size_t)(lane + 128) * plane] = v[2];
    }
    if (!p.loss_sum) continue;
    const float NEG = -__builtin_inff();
    float m = fmaxf(fmaxf(c_ok[0] ? v[0] : NEG, c_ok[1] ? v[1] : NEG), c_ok[2] ? v[2] : NEG);
    m = wave_max(m);
    float e[3];
    e[0] = c_ok[0] ? __expf(v[0] - m) : 0.f;
    e[1] = c_ok[1] ? __expf(v[1] - m) : 0.f;
    e[2] = c_ok[2] ? __expf(v[2] - m) : 0.f;
    const float ssum = wave_sum(e[0] + e[1] + e[2]);
    const bool counted = label >= 0;
    if (!counted) continue;
    const int lsel = label >> 6, llane = label & 63;
    const float vsel = lsel == 0 ? v[0] : (lsel == 1 ? v[1] : v[2]);
    const float logit_l = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vsel), llane));
    loss_acc += (m + __logf(ssum)) - logit_l;
    cnt_acc += 1.f;
    if (p.dlo) {
      const float inv = p.gscale / ssum;
      float gch[3];
      gch[0] = e[0] * inv - ((lsel == 0 && lane == llane) ? p.gscale : 0.f);
      gch[1] = e[1] * inv - ((lsel == 1 && lane == llane) ? p.gscale : 0.f);
      gch[2] = e[2] * inv - ((lsel == 2 && lane == llane) ? p.gscale : 0.f);
#pragma unroll
      for (int i = 0; i < NTAP; ++i) {
        float* rowp = s_dlo + ((ay.idx[i] - fy0) * FW - fx0) * Cp + lane;
#pragma unroll
        for (int j = 0; j < NTAP; ++j) {
          const float wgt = ay.w[i] * ax.w[j];
          float* cp = rowp + ax.idx[j] * Cp;
          if (c_ok[0]) lds_add(cp, wgt * gch[0]);
          if (c_ok[1]) lds_add(cp + 64, wgt * gch[1]);
          if (c_ok[2]) lds_add(cp + 128, wgt * gch[2]);
        }
      }
    }
  }

  __shared__ float s_red[HEAD_THREADS / 64][2];   // one pair of atomics per block (same-address atomics serialise in L2)
  if (lane == 0) { s_red[wid][0] = loss_acc; s_red[wid][1] = cnt_acc; }
  __syncthreads();
  if (p.loss_sum && tid == 0) {
    float l = 0.f, c = 0.f;
#pragma unroll
    for (int w = 0; w < HEAD_THREADS / 64; ++w) { l += s_red[w][0]; c += s_red[w][1]; }
    if (c > 0.f) {
      atomicAdd(p.loss_sum, l);
      atomicAdd(p.loss_sum + 1, c);
    }
  }
  if (p.dlo) {
    for (int i = tid; i < fsize; i += HEAD_THREADS) {
      const int cell = i / Cp, c = i % Cp;
      if (c >= p.C) continue;
      const int fy = cell / FW, fx = cell % FW;
      atomicAdd(p.dlo + (((size_t)b * p.h + fy0 + fy) * p.w + fx0 + fx) * p.ld + c, s_dlo[i]);
    }
  }
}

// ---- fast path: S == 4, bicubic (the headline configuration) or bilinear (config-5 score map, AuxiliaryLoss) ----
// For S = 4 the output pixels Y in [4a+2, 4a+6) share one tap row set {a-1..a+2} and differ only in the
// fractional weights t in {1/8, 3/8, 5/8, 7/8}.  Tiles are shifted by 2 pixels so they hold exactly 4x4 such
// groups; a wave loads a group's 4x4 low-res cells into REGISTERS once (48 values per lane, lanes = channels),
// evaluates its 16 pixels with separable row/column mixes, accumulates the gradient for the 16 cells in
// registers, and touches LDS only for the final 48 adds per group (16x fewer LDS reads / atomics than the
// generic kernel).
// Bilinear uses the same grouping with 2 taps {a, a+1} and weights {1-t, t}; torch's clamp of the source
// coordinate at 0 equals clamping the tap indices because the two clamped taps then coincide.
template <int MODE>
__global__ __launch_bounds__(HEAD_THREADS) void head_ce_s4_kernel(HeadArgs p) {
  constexpr int NT = (MODE == LC2IS_INTERP_BICUBIC) ? 4 : 2;   // taps per axis
  constexpr int OFF = (MODE == LC2IS_INTERP_BICUBIC) ? 1 : 0;  // first tap = a - OFF
  constexpr int F4 = 4 + NT - 1;                               // footprint edge: 4 groups + NT - 1
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int tiles_x = (p.W + 2 + HT - 1) / HT, tiles_y = (p.H + 2 + HT - 1) / HT;
  const int b = blockIdx.x / (tiles_x * tiles_y);
  const int tyi = (blockIdx.x / tiles_x) % tiles_y, txi = blockIdx.x % tiles_x;
  const int a0 = 4 * tyi - 1, b0 = 4 * txi - 1;  // "floor" lo index of the tile's first group row / column
  const int Cp = p.ld;
  float* s_lo = (float*)smem;
  float* s_dlo = s_lo + F4 * F4 * Cp;
  const int fsize = F4 * F4 * Cp;
  for (int i = tid * 4; i < fsize; i += HEAD_THREADS * 4) {
    const int cell = i / Cp, c = i % Cp;
    int ry = a0 - OFF + cell / F4, rx = b0 - OFF + cell % F4;
    ry = ry < 0 ? 0 : (ry > p.h - 1 ? p.h - 1 : ry);
    rx = rx < 0 ? 0 : (rx > p.w - 1 ? p.w - 1 : rx);
    *reinterpret_cast<float4*>(s_lo + i) =
        *reinterpret_cast<const float4*>(p.lo + (((size_t)b * p.h + ry) * p.w + rx) * p.ld + c);
    if (p.dlo) *reinterpret_cast<float4*>(s_dlo + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // weights of the four phases (identical for rows and columns)
  float wt[4][NT];
#pragma unroll
  for (int ph = 0; ph < 4; ++ph) {
    const float t = 0.125f + 0.25f * (float)ph;
    if constexpr (NT == 4) {
      wt[ph][0] = cubic2(t + 1.f); wt[ph][1] = cubic1(t); wt[ph][2] = cubic1(1.f - t); wt[ph][3] = cubic2(2.f - t);
    } else {
      wt[ph][0] = 1.f - t; wt[ph][1] = t;
    }
  }
  __syncthreads();

  const bool c_ok[3] = {lane < p.C, lane + 64 < p.C, lane + 128 < p.C};
  const float NEG = -__builtin_inff();
  // labels of this wave's 32 pixels: lane -> (group lane>>4, py (lane>>2)&3, px lane&3)
  int my_label = -1;
  {
    const int gi = 2 * wid + ((lane >> 4) & 1), gy = gi >> 2, gx = gi & 3;
    const int Y = 4 * (a0 + gy) + 2 + ((lane >> 2) & 3), X = 4 * (b0 + gx) + 2 + (lane & 3);
    if (lane < 32 && p.labels && Y >= 0 && X >= 0 && Y < p.H && X < p.W) {
      const int64_t lab64 = p.labels[((size_t)b * p.H + Y) * p.W + X];
      my_label = (lab64 == (int64_t)p.ignore_index || lab64 < 0 || lab64 >= p.C) ? -1 : (int)lab64;
    }
  }
  float loss_acc = 0.f, cnt_acc = 0.f;

#pragma unroll 1
  for (int g2 = 0; g2 < 2; ++g2) {
    const int gi = 2 * wid + g2, gy = gi >> 2, gx = gi & 3;
    const int Yb = 4 * (a0 + gy) + 2, Xb = 4 * (b0 + gx) + 2;
    if (Yb >= p.H || Xb >= p.W || Yb + 3 < 0 || Xb + 3 < 0) continue;  // wave-uniform
    float v[NT][NT][3], dacc[NT][NT][3];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float* cp = s_lo + ((gy + i) * F4 + gx + j) * Cp + lane;
        v[i][j][0] = cp[0]; v[i][j][1] = cp[64]; v[i][j][2] = cp[128];
        dacc[i][j][0] = 0.f; dacc[i][j][1] = 0.f; dacc[i][j][2] = 0.f;
      }
#pragma unroll
    for (int py = 0; py < 4; ++py) {
      const int Y = Yb + py;
      float r[NT][3], tq[NT][3];
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          float acc_r = 0.f;
#pragma unroll
          for (int i = 0; i < NT; ++i) acc_r += wt[py][i] * v[i][j][k];
          r[j][k] = acc_r;
          tq[j][k] = 0.f;
        }
      if (Y >= 0 && Y < p.H) {
#pragma unroll
        for (int px = 0; px < 4; ++px) {
          const int X = Xb + px;
          if (X < 0 || X >= p.W) continue;  // wave-uniform
          float lg[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            float acc_l = 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) acc_l += wt[px][j] * r[j][k];
            lg[k] = acc_l;
          }
          if (p.hi_out) {
            const size_t plane = (size_t)p.H * p.W;
            float* o = p.hi_out + ((size_t)b * p.C) * plane + (size_t)Y * p.W + X;
            if (c_ok[0]) o[(size_t)lane * plane] = lg[0];
            if (c_ok[1]) o[(size_t)(lane + 64) * plane] = lg[1];
            if (c_ok[2]) o[(size_t)(lane + 128) * plane] = lg[2];
          }
          if (!p.loss_sum) continue;
          const int label = __builtin_amdgcn_readlane(my_label, 16 * g2 + 4 * py + px);   // wave-uniform index: v_readlane
          float m = fmaxf(fmaxf(c_ok[0] ? lg[0] : NEG, c_ok[1] ? lg[1] : NEG), c_ok[2] ? lg[2] : NEG);
          m = wave_max(m);
          float e[3];
          e[0] = c_ok[0] ? __expf(lg[0] - m) : 0.f;
          e[1] = c_ok[1] ? __expf(lg[1] - m) : 0.f;
          e[2] = c_ok[2] ? __expf(lg[2] - m) : 0.f;
          const float ssum = wave_sum(e[0] + e[1] + e[2]);
          if (label < 0) continue;
          const int lsel = label >> 6, llane = label & 63;
          const float vsel = lsel == 0 ? lg[0] : (lsel == 1 ? lg[1] : lg[2]);
          loss_acc += (m + __logf(ssum)) - __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vsel), llane));
          cnt_acc += 1.f;
          if (p.dlo) {
            const float inv = p.gscale / ssum;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const float gk = e[k] * inv - ((lsel == k && lane == llane) ? p.gscale : 0.f);
#pragma unroll
              for (int j = 0; j < NT; ++j) tq[j][k] += wt[px][j] * gk;
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int k = 0; k < 3; ++k) dacc[i][j][k] += wt[py][i] * tq[j][k];
    }
    if (p.dlo) {
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          float* cp = s_dlo + ((gy + i) * F4 + gx + j) * Cp + lane;
          if (c_ok[0]) lds_add(cp, dacc[i][j][0]);
          if (c_ok[1]) lds_add(cp + 64, dacc[i][j][1]);
          if (c_ok[2]) lds_add(cp + 128, dacc[i][j][2]);
        }
    }
  }

  // one pair of atomics per BLOCK: thousands of waves adding to the same two floats serialise in one L2 channel
  // (that, not the arithmetic, was 80 % of this kernel's time)
  __shared__ float s_red[HEAD_THREADS / 64][2];
  if (lane == 0) { s_red[wid][0] = loss_acc; s_red[wid][1] = cnt_acc; }
  __syncthreads();
  if (p.loss_sum && tid == 0) {
    float l = 0.f, c = 0.f;
#pragma unroll
    for (int w = 0; w < HEAD_THREADS / 64; ++w) { l += s_red[w][0]; c += s_red[w][1]; }
    if (c > 0.f) {
      atomicAdd(p.loss_sum, l);
      atomicAdd(p.loss_sum + 1, c);
    }
  }
  if (p.dlo) {
    for (int i = tid; i < fsize; i += HEAD_THREADS) {
      const int cell = i / Cp, c = i % Cp;
      if (c >= p.C) continue;
      int ry = a0 - OFF + cell / F4, rx = b0 - OFF + cell % F4;
      ry = ry < 0 ? 0 : (ry > p.h - 1 ? p.h - 1 : ry);
      rx = rx < 0 ? 0 : (rx > p.w - 1 ? p.w - 1 : rx);
      const float val = s_dlo[i];
      if (val != 0.f) atomicAdd(p.dlo + (((size_t)b * p.h + ry) * p.w + rx) * p.ld + c, val);
    }
  }
}

// ---- generic pieces for the drop-in (unfused) path -----------------------------------------------------
// softmax cross-entropy over NCHW fp32 logits: per-pixel lse + loss; backward writes dlogits NCHW.
__global__ __launch_bounds__(256) void ce_nchw_fwd_kernel(const float* __restrict__ logits,
                                                           const int64_t* __restrict__ labels, float* lse,
                                                           float* loss_sum, int B, int C, size_t HW,
                                                           long ignore_index) {
  const size_t total = (size_t)B * HW;
  float lacc = 0.f, cacc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t b = i / HW, px = i % HW;
    const float* base = logits + b * C * HW + px;
    float m = -__builtin_inff();
    for (int c = 0; c < C; ++c) m = fmaxf(m, base[(size_t)c * HW]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += __expf(base[(size_t)c * HW] - m);
    const float l = m + __logf(s);
    if (lse) lse[i] = l;
    const long lab = (long)labels[i];
    if (lab != ignore_index && lab >= 0 && lab < C) {
      lacc += l - base[(size_t)lab * HW];
      cacc += 1.f;
    }
  }
  lacc = wave_sum(lacc);
  cacc = wave_sum(cacc);
  if ((threadIdx.x & 63) == 0 && cacc > 0.f) {
    atomicAdd(loss_sum, lacc);
    atomicAdd(loss_sum + 1, cacc);
  }
}

__global__ __launch_bounds__(256) void ce_nchw_bwd_kernel(const float* __restrict__ logits,
                                                           const int64_t* __restrict__ labels,
                                                           const float* __restrict__ lse,
                                                           const float* __restrict__ gscale_dev, float gscale,
                                                           float* dlogits, int B, int C, size_t HW,
                                                           long ignore_index) {
  const size_t total = (size_t)B * HW;
  const float gs = gscale * (gscale_dev ? *gscale_dev : 1.f);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t b = i / HW, px = i % HW;
    const float* base = logits + b * C * HW + px;
    float* dbase = dlogits + b * C * HW + px;
    const long lab = (long)labels[i];
    const bool counted = lab != ignore_index && lab >= 0 && lab < C;
    const float l = lse[i];
    for (int c = 0; c < C; ++c) {
      float g = 0.f;
      if (counted) g = gs * (__expf(base[(size_t)c * HW] - l) - (c == lab ? 1.f : 0.f));
      dbase[(size_t)c * HW] = g;
    }
  }
}

// Transposed upsample for the unfused path: dlo[b,y,x,c] = sum_{Y,X} wy(Y,y) wx(X,x) dhi[b,c,Y,X].
// grid (h, C, B), threads over x; each thread scans the <= 5S x 5S window of output pixels that can touch it.
template <int MODE>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ dhi, float* dlo, int ld,
                                                            int h, int w, int C, int S) {
  const int y = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
  const int H = h * S, W = w * S;
  const float inv_scale = 1.f / (float)S;
  const float* plane = dhi + ((size_t)b * C + c) * H * W;
  for (int x = threadIdx.x; x < w; x += 256) {
    float acc = 0.f;
    const int Y0 = max(0, S * (y - 2)), Y1 = min(H, S * (y + 3));
    const int X0 = max(0, S * (x - 2)), X1 = min(W, S * (x + 3));
    for (int Y = Y0; Y < Y1; ++Y) {
      const Taps ty = make_taps(Y, inv_scale, h, MODE);
      float wy = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) wy += (ty.idx[k] == y) ? ty.w[k] : 0.f;
      if (wy == 0.f) continue;
      float racc = 0.f;
      for (int X = X0; X < X1; ++X) {
        const Taps tx = make_taps(X, inv_scale, w, MODE);
        float wx = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) wx += (tx.idx[k] == x) ? tx.w[k] : 0.f;
        racc += wx * plane[(size_t)Y * W + X];
      }
      acc += wy * racc;
    }
    dlo[(((size_t)b * h + y) * w + x) * ld + c] = acc;
  }
}

}  // namespace

extern "C" int lc2is_upsample_bwd_nchw(const float* dhi, float* dlo, int ld, int B, int h, int w, int C, int S,
                                       int mode, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!dhi || !dlo) return LC2IS_ERR_NULL;
  if (B <= 0 || h <= 0 || w <= 0 || C <= 0 || ld < C || S < 1) return LC2IS_ERR_SHAPE;
  if (mode == LC2IS_INTERP_BICUBIC)
    hipLaunchKernelGGL(upsample_bwd_kernel<LC2IS_INTERP_BICUBIC>, dim3(h, C, B), dim3(256), 0, stream, dhi, dlo,
                       ld, h, w, C, S);
  else if (mode == LC2IS_INTERP_BILINEAR)
    hipLaunchKernelGGL(upsample_bwd_kernel<LC2IS_INTERP_BILINEAR>, dim3(h, C, B), dim3(256), 0, stream, dhi, dlo,
                       ld, h, w, C, S);
  else
    return LC2IS_ERR_UNSUPPORTED;
  return lc2is_check_launch();
}

extern "C" int lc2is_head_upsample_ce(const float* scores_lo, int ld, const int64_t* labels, float* dscores_lo,
                                      float* scores_hi, float* loss_sum, int B, int h, int w, int C, int S,
                                      int mode, long ignore_index, float grad_scale, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!scores_lo) return LC2IS_ERR_NULL;
  if (!scores_hi && !loss_sum) return LC2IS_ERR_NULL;
  if ((loss_sum || dscores_lo) && !labels) return LC2IS_ERR_NULL;
  if (dscores_lo && !loss_sum) return LC2IS_ERR_NULL;
  if (B <= 0 || h <= 0 || w <= 0 || C <= 0 || C > CMAX || ld < C || ld > CMAX || ld % 64) return LC2IS_ERR_SHAPE;
  if (S < 4 || (HT % S != 0 && S % HT != 0)) return LC2IS_ERR_UNSUPPORTED;
  if (mode != LC2IS_INTERP_BICUBIC && mode != LC2IS_INTERP_BILINEAR) return LC2IS_ERR_UNSUPPORTED;
  const int H = h * S, W = w * S;
  HeadArgs a{scores_lo, ld, labels, dscores_lo, scores_hi, loss_sum, B, h, w, H, W, C, S, mode, ignore_index,
             grad_scale};
  const int lds_bytes = 2 * FMAX * FMAX * ld * (int)sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    const int mx = 2 * FMAX * FMAX * CMAX * (int)sizeof(float);
    if (hipFuncSetAttribute((const void*)head_ce_kernel<LC2IS_INTERP_BICUBIC>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
        hipFuncSetAttribute((const void*)head_ce_kernel<LC2IS_INTERP_BILINEAR>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess)
      return LC2IS_ERR_LAUNCH;
    attr_set = true;
  }
  if (S == 4) {
    static bool attr4 = false;
    const int f4 = (mode == LC2IS_INTERP_BICUBIC) ? 7 : 5;
    const int lds4 = 2 * f4 * f4 * ld * (int)sizeof(float);
    if (!attr4) {
      const int mx4 = 2 * 7 * 7 * CMAX * (int)sizeof(float);
      if (hipFuncSetAttribute((const void*)head_ce_s4_kernel<LC2IS_INTERP_BICUBIC>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, mx4) != hipSuccess ||
          hipFuncSetAttribute((const void*)head_ce_s4_kernel<LC2IS_INTERP_BILINEAR>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, mx4) != hipSuccess)
        return LC2IS_ERR_LAUNCH;
      attr4 = true;
    }
    const int t4 = ((H + 2 + HT - 1) / HT) * ((W + 2 + HT - 1) / HT);
    if (mode == LC2IS_INTERP_BICUBIC)
      hipLaunchKernelGGL(head_ce_s4_kernel<LC2IS_INTERP_BICUBIC>, dim3(B * t4), dim3(HEAD_THREADS), lds4, stream, a);
    else
      hipLaunchKernelGGL(head_ce_s4_kernel<LC2IS_INTERP_BILINEAR>, dim3(B * t4), dim3(HEAD_THREADS), lds4, stream, a);
    return lc2is_check_launch();
  }
  const int tiles = ((H + HT - 1) / HT) * ((W + HT - 1) / HT);
  if (mode == LC2IS_INTERP_BICUBIC)
    hipLaunchKernelGGL(head_ce_kernel<LC2IS_INTERP_BICUBIC>, dim3(B * tiles), dim3(HEAD_THREADS), lds_bytes,
                       stream, a);
  else
    hipLaunchKernelGGL(head_ce_kernel<LC2IS_INTERP_BILINEAR>, dim3(B * tiles), dim3(HEAD_THREADS), lds_bytes,
                       stream, a);
  return lc2is_check_launch();
}

extern "C" int lc2is_ce_nchw_fwd(const float* logits, const int64_t* labels, float* lse, float* loss_sum, int B,
                                 int C, long HW, long ignore_index, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!logits || !labels || !loss_sum) return LC2IS_ERR_NULL;
  if (B <= 0 || C <= 0 || HW <= 0) return LC2IS_ERR_SHAPE;
  size_t g = ((size_t)B * HW + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(ce_nchw_fwd_kernel, dim3((int)g), dim3(256), 0, stream, logits, labels, lse, loss_sum, B, C,
                     (size_t)HW, ignore_index);
  return lc2is_check_launch();
}

extern "C" int lc2is_ce_nchw_bwd(const float* logits, const int64_t* labels, const float* lse,
                                 const float* grad_scale_dev, float grad_scale, float* dlogits, int B, int C,
                                 long HW, long ignore_index, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!logits || !labels || !lse || !dlogits) return LC2IS_ERR_NULL;
  if (B <= 0 || C <= 0 || HW <= 0) return LC2IS_ERR_SHAPE;
  size_t g = ((size_t)B * HW + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(ce_nchw_bwd_kernel, dim3((int)g), dim3(256), 0, stream, logits, labels, lse, grad_scale_dev,
                     grad_scale, dlogits, B, C, (size_t)HW, ignore_index);
  return lc2is_check_launch();
}
