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
// Work split: a block owns a 16x16 tile of OUTPUT pixels; the <=7x7 low-res footprint of the tile (S >= 4) is staged in LDS
// once and the gradient wrt the low-res scores is accumulated in an LDS mirror of the footprint.  S = 4 / 8 / 16 (round 5): the
// mirror leaves as the block's own SLAB of a workspace (plain stores) and a second launch sums, for every low-res cell, the
// footprint cells of the tiles that cover it in a FIXED order (tile row, footprint row, tile column, footprint column); the
// blocks' loss / count partials go the same way — no float atomics, the loss and the gradient are bitwise reproducible.
// (Rounds 1-4 flushed the mirror with fp32 atomics: the arrival order made the last bits of the step run-dependent.)
// Other S keep the atomic flush (not reproducible to the last bit; no configuration of the reference reaches them).
//  * S = 4, 8, 16 (the headline bicubic x4 head, config 5's bilinear x4 score map, AuxiliaryLoss at 32 -> 512): head_ce_grp_kernel —
//    S x S-pixel groups in units of 16 pixels as two small products on the fp32 matrix pipe, softmax in the accumulator layout,
//    waves taking turns to add their gradient tiles to the LDS mirror (no LDS atomics: they retire about one lane per three
//    cycles per CU and were 90 % of the first S = 4 kernel).
//  * other S (multiples of 16 from 32): head_ce_kernel — each wave walks 64 output pixels with the 64 lanes spread over CHANNELS (3 per lane, C <= 192),
//    wave reductions for the softmax, ds_add_f32 for the gradient scatter.
#include "common.h"
#include "lc2is_hip.h"

namespace {

constexpr int HT = 16;        // output tile edge
constexpr int FMAX = 8;       // max footprint edge for S >= 4 (16/S + 4 bicubic rows)
constexpr int HEAD_THREADS = 512;
constexpr int CMAX = 192;
constexpr int HEAD_PAD = 4;     // floats added to a footprint cell's LDS stride in the S = 4 kernel

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
  // group kernels (S = 4 / 8 / 16): per-block outputs, summed in fixed order by head_finish_kernel
  float* ws_loss;                // [blocks][2] loss / count partials
  float* ws_dlo;                 // [blocks][F4 * F4][cq] footprint mirrors
  int cq;                        // channels per slab cell: C rounded up to 4
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

// ---- S = 4 / 8 / 16, bicubic (S = 4: the headline configuration) or bilinear (config-5 score map, AuxiliaryLoss) ----
// The output pixels Y in [S a + S/2, S a + 3S/2) share one tap row set {a-1..a+2} (bilinear: {a, a+1}) and differ only in the
// fractional weight t = (ph + 1/2) / S, ph = 0..S-1; tiles are shifted by S/2 pixels so that they hold exactly (16/S)^2 such GROUPS
// of S x S pixels.  A group is worked in UNITS of 16 pixels (S = 4: the whole 4x4 group; S = 8: two rows of 8; S = 16: one row), two
// units per wave, and a unit is two small matrix products on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: an exact k-ordered fp32
// fma chain at twice the plain-VALU rate, beside the VALU instead of on it):
//   logits[16 pixels][C]  = Wm[16 pixels][NT*NT cells] x lo[cells][C]            (forward,  A = weights, B = LDS footprint)
//   dlo^T [C][cells]     += dlogits^T[C][16 pixels]    x Wm[16 pixels][cells]    (backward, A = the forward's accumulators)
// Wm[pixel (py,px)][cell (i,j)] = w(py,i) * w(px,j).  The accumulator of a 16-channel tile holds, per lane, channel
// 16t + (lane & 15) of the four pixels 4 (lane >> 4) + r of the unit: the softmax over channels is an in-lane loop over the
// tiles plus four DPP steps inside a 16-lane row, for four pixels at once, and the same registers (now dlogits) are the A
// operand of the backward product once its B operand is ordered to match (k = lane >> 4 <-> pixel 4 k + jj in product jj).
// The label's logit (for the loss) is 16 x NT*NT scalars per unit: LDS reads with one (pixel, cell quad) per lane; its one-hot
// is subtracted from dlogits in the accumulator layout.
// Bilinear's clamp of the source coordinate at 0 (torch) equals clamping the tap indices because the two clamped taps coincide.
template <int MODE, int S> __device__ __forceinline__ float tap_w(int ph, int k) {
  const float t = ((float)ph + 0.5f) * (1.f / (float)S);
  if constexpr (MODE == LC2IS_INTERP_BICUBIC)
    return k == 0 ? cubic2(t + 1.f) : (k == 1 ? cubic1(t) : (k == 2 ? cubic1(1.f - t) : cubic2(2.f - t)));
  else
    return k == 0 ? 1.f - t : t;
}

template <int CTRL> __device__ __forceinline__ float row_dpp(float v) {   // lane permutation inside each row of 16 lanes
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, row_dpp<0xB1>(v)); v = fmaxf(v, row_dpp<0x4E>(v)); v = fmaxf(v, row_dpp<0x141>(v));
  return fmaxf(v, row_dpp<0x140>(v));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += row_dpp<0xB1>(v); v += row_dpp<0x4E>(v); v += row_dpp<0x141>(v);
  return v + row_dpp<0x140>(v);
}

constexpr int head_grp_footprint(int mode, int S) { return HT / S + (mode == LC2IS_INTERP_BICUBIC ? 4 : 2) - 1; }

template <int MODE, int TN, int S>
__global__ __launch_bounds__(HEAD_THREADS, (S == 4 ? 2 : 1)) void head_ce_grp_kernel(HeadArgs p) {
  constexpr int NT = (MODE == LC2IS_INTERP_BICUBIC) ? 4 : 2;   // taps per axis
  constexpr int OFF = (MODE == LC2IS_INTERP_BICUBIC) ? 1 : 0;  // first tap = a - OFF
  constexpr int G = HT / S;                                    // groups per tile edge
  constexpr int F4 = G + NT - 1;                               // footprint edge
  constexpr int NQ = NT * NT / 4;                              // k = 4 chunks of the forward product
  constexpr int UPG = S * S / 16;                              // 16-pixel units per group (whole rows of the group: 16 / S rows each)
  constexpr bool SUMD = UPG > 1;                               // a wave's two units belong to one group: their gradient tiles are summed
  static_assert(S == 4 || S == 8 || S == 16, "group kernel: S x S pixel groups inside a 16 x 16 tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int tiles_x = (p.W + S / 2 + HT - 1) / HT, tiles_y = (p.H + S / 2 + HT - 1) / HT;
  const int b = blockIdx.x / (tiles_x * tiles_y);
  const int tyi = (blockIdx.x / tiles_x) % tiles_y, txi = blockIdx.x % tiles_x;
  const int a0 = G * tyi - 1, b0 = G * txi - 1;  // "floor" lo index of the tile's first group row / column
  const int Y0 = S * a0 + S / 2, X0 = S * b0 + S / 2;   // the tile's first output pixel
  const int Cp = p.ld;
  const int CS = Cp + HEAD_PAD;                  // cell stride in LDS (floats): the 16 cells of a group on different banks
  float* s_lo = (float*)smem;
  float* s_dlo = s_lo + F4 * F4 * CS;
  int* s_lab = (int*)(s_dlo + F4 * F4 * CS);     // [16][16] pixels of the tile: -2 outside the image, -1 not counted, else label
  const int fsize = F4 * F4 * Cp;
  for (int i = tid * 4; i < fsize; i += HEAD_THREADS * 4) {
    const int cell = i / Cp, c = i % Cp;
    int ry = a0 - OFF + cell / F4, rx = b0 - OFF + cell % F4;
    ry = ry < 0 ? 0 : (ry > p.h - 1 ? p.h - 1 : ry);
    rx = rx < 0 ? 0 : (rx > p.w - 1 ? p.w - 1 : rx);
    *reinterpret_cast<float4*>(s_lo + cell * CS + c) =
        *reinterpret_cast<const float4*>(p.lo + (((size_t)b * p.h + ry) * p.w + rx) * p.ld + c);
    if (p.dlo) *reinterpret_cast<float4*>(s_dlo + cell * CS + c) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (tid < HT * HT) {
    const int Y = Y0 + (tid >> 4), X = X0 + (tid & 15);
    int code = -2;
    if (Y >= 0 && X >= 0 && Y < p.H && X < p.W) {
      code = -1;
      if (p.labels) {
        const int64_t lab64 = p.labels[((size_t)b * p.H + Y) * p.W + X];
        if (lab64 != (int64_t)p.ignore_index && lab64 >= 0 && lab64 < p.C) code = (int)lab64;
      }
    }
    s_lab[tid] = code;
  }
  __syncthreads();

  const int m = lane & 15, kq = lane >> 4;
  const float NEG = -__builtin_inff();
  constexpr float LOG2E = 1.4426950408889634f;
  const int cellm_off = ((m / NT) * F4 + m % NT) * CS + 4 * kq;   // this lane's cell / channel quad in the transposed gradient tile
  int cell_off[NQ];   // LDS float offset of the forward cell 4q + kq, relative to the group's first cell
#pragma unroll
  for (int q = 0; q < NQ; ++q) cell_off[q] = (((4 * q + kq) / NT) * F4 + (4 * q + kq) % NT) * CS;
  // forward A: row = pixel m of the unit, k = kq -> cell 4q + kq.  Backward B of product jj: column = cell m, k = kq -> pixel 4 kq + jj
  float Af[NQ], Bb[4];
  auto make_operands = [&](int prow) {
    const int py_m = prow + m / S, px_m = m % S;
#pragma unroll
    for (int q = 0; q < NQ; ++q) Af[q] = tap_w<MODE, S>(py_m, (4 * q + kq) / NT) * tap_w<MODE, S>(px_m, (4 * q + kq) % NT);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int n = 4 * kq + jj;
      Bb[jj] = (m < NT * NT) ? tap_w<MODE, S>(prow + n / S, m / NT) * tap_w<MODE, S>(n % S, m % NT) : 0.f;
    }
  };
  if constexpr (!SUMD) make_operands(0);
  float loss_acc = 0.f, cnt_acc = 0.f;
  f32x4_t dsum[SUMD ? TN : 1];
  if constexpr (SUMD) {
#pragma unroll
    for (int t = 0; t < TN; ++t) dsum[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  bool have_d = false;
  int gbase = 0;

#pragma unroll 1
  for (int g2 = 0; g2 < 2; ++g2) {
    const int u = 2 * wid + g2, gi = u / UPG, rt = u % UPG, gy = gi / G, gx = gi % G;
    const int Yb = Y0 + S * gy, Xb = X0 + S * gx;   // the group's first output pixel
    // pixel n of the unit: (py, px) inside the group
    const int prow = rt * (16 / S);
    const bool active = !(Yb + prow >= p.H || Xb >= p.W || Yb + prow + 16 / S - 1 < 0 || Xb + S - 1 < 0);  // wave-uniform
    gbase = (gy * F4 + gx) * CS;
    f32x4_t acc[TN];   // logits, then exp, then (S = 4: in place) the transposed gradient tiles
    if constexpr (!SUMD) have_d = false;
    if (active) {
      const int py_m = prow + m / S, px_m = m % S;
      if constexpr (SUMD) make_operands(prow);   // (S = 4: one unit shape, formed once in front of the loop)
#pragma unroll
      for (int t = 0; t < TN; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const float* bp = s_lo + gbase + cell_off[q] + m;
        float bv[TN];
#pragma unroll
        for (int t = 0; t < TN; ++t) bv[t] = bp[16 * t];
#pragma unroll
        for (int t = 0; t < TN; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Af[q], bv[t], acc[t], 0, 0, 0);
      }
      // this lane's four accumulator pixels 4 kq .. 4 kq + 3 of the unit: one row of the tile, four consecutive columns
      const int py4 = prow + (4 * kq) / S, px4 = (4 * kq) % S;
      const i32x4_t lab4 = *reinterpret_cast<const i32x4_t*>(s_lab + (S * gy + py4) * 16 + S * gx + px4);
      if (p.hi_out) {
        const size_t plane = (size_t)p.H * p.W;
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          const int ch = 16 * t + m;
          if (ch < p.C) {
            float* o = p.hi_out + ((size_t)b * p.C + ch) * plane + (size_t)(Yb + py4) * p.W + Xb + px4;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (lab4[r] != -2) o[r] = acc[t][r];
          }
        }
      }
      if (p.loss_sum) {
        // the label's logit: this lane owns pixel m x cells {4q + kq}
        const int lab_m = s_lab[(S * gy + py_m) * 16 + S * gx + px_m];
        if (lab_m >= 0) {
#pragma unroll
          for (int q = 0; q < NQ; ++q) loss_acc -= Af[q] * s_lo[gbase + cell_off[q] + lab_m];
        }
        // softmax over the channels of four pixels at once
        float mx[4] = {NEG, NEG, NEG, NEG}, sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          if (16 * t + 16 > p.C) {   // uniform: a tile with channels past C
            if (16 * t + m >= p.C) acc[t] = f32x4_t{NEG, NEG, NEG, NEG};
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) mx[r] = fmaxf(mx[r], acc[t][r]);
        }
        float mxl[4];   // -max * log2(e): exp(x - max) = exp2(fma(x, log2(e), mxl))
#pragma unroll
        for (int r = 0; r < 4; ++r) { mx[r] = row16_max(mx[r]); mxl[r] = -mx[r] * LOG2E; }
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[t][r], LOG2E, mxl[r]));
            acc[t][r] = e;
            sum[r] += e;
          }
        float inv[4];
        int dl[4];   // label - lane's channel offset: the one-hot sits in tile t where dl == 16 t
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sum[r] = row16_sum(sum[r]);
          const bool counted = lab4[r] >= 0;
          inv[r] = counted ? p.gscale / sum[r] : 0.f;
          dl[r] = counted ? lab4[r] - m : -1;
          if (counted && m == 0) {
            loss_acc += mx[r] + __logf(sum[r]);
            cnt_acc += 1.f;
          }
        }
        if (p.dlo) {
          have_d = true;
#pragma unroll
          for (int t = 0; t < TN; ++t) {
            f32x4_t gv;
#pragma unroll
            for (int r = 0; r < 4; ++r) gv[r] = acc[t][r] * inv[r] - (dl[r] == 16 * t ? p.gscale : 0.f);
            f32x4_t d = {0.f, 0.f, 0.f, 0.f};
            if constexpr (SUMD) d = dsum[t];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) d = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[jj], Bb[jj], d, 0, 0, 0);
            if constexpr (SUMD) dsum[t] = d; else acc[t] = d;   // d[r] = channel 16 t + 4 kq + r of cell m
          }
        }
      }
    }
    // The groups of a tile overlap in every footprint cell (bicubic), and LDS float atomics retire about one LANE per three
    // cycles per CU — 2 800 lane-adds per group made them 90 % of the first S = 4 kernel.  The waves take turns instead: plain
    // 16-byte read-add-write of the transposed tiles (one cell per lane, four consecutive channels, conflict-free under the padded
    // stride).  S = 4 bilinear footprints are 2 x 2 cells: the groups a pass runs on waves of equal (wid >> 1) & 1 sit two cells
    // apart in both directions, so two turns per pass do.  S >= 8: a wave's two units are one group's, one round of turns per tile.
    if (p.dlo && (!SUMD || g2 == 1)) {
      constexpr int NTURN = (S == 4 && MODE != LC2IS_INTERP_BICUBIC) ? 2 : HEAD_THREADS / 64;
      const int my_turn = (S == 4 && MODE != LC2IS_INTERP_BICUBIC) ? ((wid >> 1) & 1) : wid;
      for (int turn = 0; turn < NTURN; ++turn) {
        __syncthreads();
        if (turn == my_turn && have_d && m < NT * NT) {
          float* dp = s_dlo + gbase + cellm_off;
#pragma unroll
          for (int t = 0; t < TN; ++t) {
            f32x4_t v = *reinterpret_cast<f32x4_t*>(dp + 16 * t);
            if constexpr (SUMD) v += dsum[t]; else v += acc[t];
            *reinterpret_cast<f32x4_t*>(dp + 16 * t) = v;
          }
        }
      }
    }
  }

  // one pair of atomics per BLOCK: thousands of waves adding to the same two floats serialise in one L2 channel
  __shared__ float s_red[HEAD_THREADS / 64][2];
  loss_acc = wave_sum(loss_acc);
  cnt_acc = wave_sum(cnt_acc);
  if (lane == 0) { s_red[wid][0] = loss_acc; s_red[wid][1] = cnt_acc; }
  __syncthreads();
  if (p.loss_sum && tid == 0) {
    float l = 0.f, c = 0.f;
#pragma unroll
    for (int w = 0; w < HEAD_THREADS / 64; ++w) { l += s_red[w][0]; c += s_red[w][1]; }
    p.ws_loss[2 * (size_t)blockIdx.x] = l;
    p.ws_loss[2 * (size_t)blockIdx.x + 1] = c;
  }
  if (p.dlo) {   // the mirror leaves as this block's slab: cells in footprint order, cq channels each, 16 bytes per lane
    const int q4 = p.cq >> 2;
    float* slab = p.ws_dlo + (size_t)blockIdx.x * (F4 * F4) * p.cq;
    for (int i = tid; i < F4 * F4 * q4; i += HEAD_THREADS) {
      const int cell = i / q4, c = (i % q4) * 4;
      *reinterpret_cast<float4*>(slab + (size_t)cell * p.cq + c) = *reinterpret_cast<const float4*>(s_dlo + cell * CS + c);
    }
  }
}

// dlo[b, y, x, :] = sum of the footprint cells that map to low-res cell (y, x), over the tiles that cover it, in a fixed order;
// loss_sum = sum of the blocks' partials in block order.  One wave per low-res cell (lane = channel quad), four cells per block.
// Tile t along an axis covers the unclamped low-res indices [G t - 1 - OFF, G t - 1 - OFF + F4); indices below 0 / above n - 1 are
// the clamped taps of the image border and belong to cell 0 / n - 1.
template <int MODE, int S>
__global__ __launch_bounds__(256) void head_finish_kernel(HeadArgs p, int nblk_main) {
  constexpr int NT = (MODE == LC2IS_INTERP_BICUBIC) ? 4 : 2;
  constexpr int OFF = (MODE == LC2IS_INTERP_BICUBIC) ? 1 : 0;
  constexpr int G = HT / S;
  constexpr int F4 = G + NT - 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (p.loss_sum && blockIdx.x == 0 && wave == 0) {
    float l = 0.f, c = 0.f;
    for (int k = lane; k < nblk_main; k += 64) { l += p.ws_loss[2 * (size_t)k]; c += p.ws_loss[2 * (size_t)k + 1]; }
    l = wave_sum(l);
    c = wave_sum(c);
    if (lane == 0) { p.loss_sum[0] = l; p.loss_sum[1] = c; }
  }
  if (!p.dlo) return;
  const long ncell = (long)p.B * p.h * p.w;
  const long cid = (long)blockIdx.x * 4 + wave;
  if (cid >= ncell) return;
  const int x = (int)(cid % p.w), y = (int)((cid / p.w) % p.h), b = (int)(cid / ((long)p.w * p.h));
  const int tiles_x = (p.W + S / 2 + HT - 1) / HT, tiles_y = (p.H + S / 2 + HT - 1) / HT;
  auto floordiv = [](int a, int d) { return a >= 0 ? a / d : -((-a + d - 1) / d); };
  const int ty_lo = max(0, floordiv(y + 1 + OFF - (F4 - 1), G)), ty_hi = (y == p.h - 1) ? tiles_y - 1 : min(tiles_y - 1, (y + 1 + OFF) / G);
  const int tx_lo = max(0, floordiv(x + 1 + OFF - (F4 - 1), G)), tx_hi = (x == p.w - 1) ? tiles_x - 1 : min(tiles_x - 1, (x + 1 + OFF) / G);
  const int q4 = p.cq >> 2;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int ty = ty_lo; ty <= ty_hi; ++ty) {
    const int by = G * ty - 1 - OFF;
    const int fy_lo = max(0, y == 0 ? 0 : y - by), fy_hi = min(F4 - 1, y == p.h - 1 ? F4 - 1 : y - by);
    for (int fy = fy_lo; fy <= fy_hi; ++fy)
      for (int tx = tx_lo; tx <= tx_hi; ++tx) {
        const int bx = G * tx - 1 - OFF;
        const int fx_lo = max(0, x == 0 ? 0 : x - bx), fx_hi = min(F4 - 1, x == p.w - 1 ? F4 - 1 : x - bx);
        const float* slab = p.ws_dlo + (((size_t)b * tiles_y + ty) * tiles_x + tx) * (F4 * F4) * p.cq;
        for (int fx = fx_lo; fx <= fx_hi; ++fx)
          if (lane < q4) {
            const float4 v = *reinterpret_cast<const float4*>(slab + (size_t)(fy * F4 + fx) * p.cq + 4 * lane);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
          }
      }
  }
  if (4 * lane < p.ld) {   // every channel of the row is written (zeros past cq): the caller need not clear the buffer
    float* o = p.dlo + (((size_t)b * p.h + y) * p.w + x) * p.ld + 4 * lane;
    if (4 * lane + 0 >= p.C) acc.x = 0.f;
    if (4 * lane + 1 >= p.C) acc.y = 0.f;
    if (4 * lane + 2 >= p.C) acc.z = 0.f;
    if (4 * lane + 3 >= p.C) acc.w = 0.f;
    *reinterpret_cast<float4*>(o) = acc;
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

namespace {
struct HeadGrpPlan { int f4, t4, cq; size_t loss_bytes, dlo_bytes; };
// the S = 4 / 8 / 16 group kernels: blocks per image, slab cell width, workspace split (loss partials first, 256-byte aligned slabs)
HeadGrpPlan head_grp_plan(int B, int h, int w, int C, int S, int mode, bool want_grad) {
  HeadGrpPlan g;
  g.f4 = head_grp_footprint(mode, S);
  const int H = h * S, W = w * S;
  g.t4 = ((H + S / 2 + HT - 1) / HT) * ((W + S / 2 + HT - 1) / HT);
  g.cq = (C + 3) & ~3;
  g.loss_bytes = ((size_t)B * g.t4 * 2 * sizeof(float) + 255) & ~(size_t)255;
  g.dlo_bytes = want_grad ? (size_t)B * g.t4 * g.f4 * g.f4 * g.cq * sizeof(float) : 0;
  return g;
}
}  // namespace

extern "C" size_t lc2is_head_upsample_ce_workspace_bytes(int B, int h, int w, int C, int S, int mode, int want_grad) {
  if (B <= 0 || h <= 0 || w <= 0 || C <= 0 || !(S == 4 || S == 8 || S == 16)) return 0;   // other S: the atomic path, no workspace
  if (mode != LC2IS_INTERP_BICUBIC && mode != LC2IS_INTERP_BILINEAR) return 0;
  const HeadGrpPlan g = head_grp_plan(B, h, w, C, S, mode, want_grad != 0);
  return g.loss_bytes + g.dlo_bytes;
}

extern "C" int lc2is_head_upsample_ce(const float* scores_lo, int ld, const int64_t* labels, float* dscores_lo,
                                      float* scores_hi, float* loss_sum, int B, int h, int w, int C, int S,
                                      int mode, long ignore_index, float grad_scale, void* workspace,
                                      size_t workspace_bytes, lc2is_stream_t stream_) {
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
             grad_scale, nullptr, nullptr, 0};
  const int lds_bytes = 2 * FMAX * FMAX * ld * (int)sizeof(float);
  static DevOnce attr_set;
  if (attr_set.need()) {
    const int mx = 2 * FMAX * FMAX * CMAX * (int)sizeof(float);
    if (hipFuncSetAttribute((const void*)head_ce_kernel<LC2IS_INTERP_BICUBIC>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
        hipFuncSetAttribute((const void*)head_ce_kernel<LC2IS_INTERP_BILINEAR>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess)
      return LC2IS_ERR_LAUNCH;
    attr_set.done();
  }
  if (S == 4 || S == 8 || S == 16) {
    // channel tiles of 16 the kernel runs (tiles past C are masked): the smallest instantiation that covers C inside the row stride
    const int nt = (C + 15) / 16;
    const int tn = nt <= 4 ? 4 : (nt <= 8 ? 8 : (nt <= 10 ? 10 : 12));
    const HeadGrpPlan gp = head_grp_plan(B, h, w, C, S, mode, dscores_lo != nullptr);
    const int f4 = gp.f4, t4 = gp.t4;
    const int lds4 = 2 * f4 * f4 * (ld + HEAD_PAD) * (int)sizeof(float) + HT * HT * (int)sizeof(int);
    if (loss_sum) {   // loss / gradient leave through per-block partials and slabs: a workspace is part of the call
      if (!workspace || ((size_t)workspace & 15) || workspace_bytes < gp.loss_bytes + gp.dlo_bytes) return LC2IS_ERR_WORKSPACE;
      a.ws_loss = (float*)workspace;
      a.ws_dlo = dscores_lo ? (float*)((char*)workspace + gp.loss_bytes) : nullptr;
      a.cq = gp.cq;
    }
#define LC2IS_HEAD_GRP(MODE_, TN_, S_)                                                                                       \
  do {                                                                                                                      \
    static DevOnce attr;                                                                                               \
    if (attr.need()) {                                                                                                            \
      if (hipFuncSetAttribute((const void*)head_ce_grp_kernel<MODE_, TN_, S_>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                              2 * 7 * 7 * (CMAX + HEAD_PAD) * (int)sizeof(float) + HT * HT * (int)sizeof(int)) != hipSuccess) \
        return LC2IS_ERR_LAUNCH;                                                                                            \
      attr.done();                                                                                                          \
    }                                                                                                                       \
    hipLaunchKernelGGL((head_ce_grp_kernel<MODE_, TN_, S_>), dim3(B * t4), dim3(HEAD_THREADS), lds4, stream, a);            \
  } while (0)
#define LC2IS_HEAD_GRP_TN(MODE_, S_)                                                                                         \
  do {                                                                                                                      \
    if (tn == 4) LC2IS_HEAD_GRP(MODE_, 4, S_); else if (tn == 8) LC2IS_HEAD_GRP(MODE_, 8, S_);                               \
    else if (tn == 10) LC2IS_HEAD_GRP(MODE_, 10, S_); else LC2IS_HEAD_GRP(MODE_, 12, S_);                                    \
  } while (0)
#define LC2IS_HEAD_GRP_S(MODE_)                                                                                              \
  do {                                                                                                                      \
    if (S == 4) LC2IS_HEAD_GRP_TN(MODE_, 4); else if (S == 8) LC2IS_HEAD_GRP_TN(MODE_, 8); else LC2IS_HEAD_GRP_TN(MODE_, 16); \
  } while (0)
    if (mode == LC2IS_INTERP_BICUBIC) LC2IS_HEAD_GRP_S(LC2IS_INTERP_BICUBIC);
    else LC2IS_HEAD_GRP_S(LC2IS_INTERP_BILINEAR);
#undef LC2IS_HEAD_GRP_S
#undef LC2IS_HEAD_GRP_TN
#undef LC2IS_HEAD_GRP
    int rc = lc2is_check_launch();
    if (rc || !loss_sum) return rc;
    // second launch: fixed-order sums of the slabs (one wave per low-res cell) and of the loss partials
    const long ncell = dscores_lo ? (long)B * h * w : 1;
    const int fgrid = (int)((ncell + 3) / 4);
#define LC2IS_HEAD_FIN(MODE_, S_) hipLaunchKernelGGL((head_finish_kernel<MODE_, S_>), dim3(fgrid), dim3(256), 0, stream, a, B * t4)
    if (mode == LC2IS_INTERP_BICUBIC) { if (S == 4) LC2IS_HEAD_FIN(LC2IS_INTERP_BICUBIC, 4); else if (S == 8) LC2IS_HEAD_FIN(LC2IS_INTERP_BICUBIC, 8); else LC2IS_HEAD_FIN(LC2IS_INTERP_BICUBIC, 16); }
    else { if (S == 4) LC2IS_HEAD_FIN(LC2IS_INTERP_BILINEAR, 4); else if (S == 8) LC2IS_HEAD_FIN(LC2IS_INTERP_BILINEAR, 8); else LC2IS_HEAD_FIN(LC2IS_INTERP_BILINEAR, 16); }
#undef LC2IS_HEAD_FIN
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
