// LayerNorm forward / backward over the fp32 residual stream (gfx950).
// One wave64 per row; a row of C <= 2048 floats lives in registers as up to 8 float4 per lane
// (16-byte coalesced loads, 1 KiB per wave instruction), statistics by wave shuffles — HBM-bound,
// no LDS needed on the forward.  Backward folds the residual-path gradient add into the same pass and
// writes the new gradient stream both as fp32 (residual chain) and bf16 (operand of the next dgrad /
// wgrad MFMA GEMMs), and reduces dgamma/dbeta deterministically via per-block partials.
// replaces: nn.LayerNorm at hf:modeling_clip.py:362-383 (layer_norm1/2), :604-607 (pre_layrnorm),
//           final_layer_norm (hf CLIPTextModel), norm1-3 of model/decoder.py:9 DecoderLayer.
#include "common.h"
#include "lc2is_hip.h"
#include <cstdlib>

namespace {

#ifndef LN_BWD_OCC
#define LN_BWD_OCC 4   // waves per SIMD the backward is compiled for when a lane holds <= 3 float4 groups (C <= 768: <= 128 VGPRs, measured against 3); wider rows get 3 (C <= 1024) or 2: at 4 they spilled 84 .. 368 VGPRs (ViT-L: 196 us per launch)
#endif
constexpr int LN_MAXV = 8;  // float4 per lane -> C <= 2048 (kernels are instantiated for NV = 1,2,3,4,6,8)

__device__ __forceinline__ float4 load_bf16x4(const bf16_t* p) {   // four consecutive bf16 (8-byte aligned) -> fp32
  const uint2 pk = *reinterpret_cast<const uint2*>(p);
  return make_float4(bf16_to_f32((bf16_t)(pk.x & 0xffff)), bf16_to_f32((bf16_t)(pk.x >> 16)),
                     bf16_to_f32((bf16_t)(pk.y & 0xffff)), bf16_to_f32((bf16_t)(pk.y >> 16)));
}

// A wave walks rows blockIdx.x * 4 + wave, + 4 * gridDim.x, ... : gamma / beta are read once per wave instead of once per row and the
// next row's x is requested before the current row's reductions (round 4; worth 0.2 % of the step: 1018 vs 1016 img/s against one
// row per wave).  The arithmetic of a row is unchanged (bitwise the same outputs).
template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const bf16_t* __restrict__ xb, int ldx,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, bf16_t* y, int ldy,
                                                      float* yf, int ldyf, float* mean_out, float* rstd_out,
                                                      int M, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int C4 = C >> 2;
  float4 gm[NV], bt[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = lane + 64 * i;
    gm[i] = (c4 < C4) ? reinterpret_cast<const float4*>(gamma)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    bt[i] = (beta && c4 < C4) ? reinterpret_cast<const float4*>(beta)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int stride = gridDim.x * 4;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  auto load_row = [&](int r, float4 (&dst)[NV]) __attribute__((always_inline)) {
    if (xb) {   // bf16 residual stream (round 5): 8 bytes per lane and group, widened on the way in
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c4 = lane + 64 * i;
        dst[i] = (c4 < C4) ? load_bf16x4(xb + (size_t)r * ldx + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      return;
    }
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)r * ldx);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c4 = lane + 64 * i;
      dst[i] = (c4 < C4) ? xr[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  float4 v[NV], nx[NV];
  load_row(row, v);
  for (; row < M; row += stride) {
    const bool more = row + stride < M;
    if (more) load_row(row + stride, nx);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < C4) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < C4) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    if (lane == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < C4) {
        float4 o;
        o.x = (v[i].x - mean) * rstd * gm[i].x;
        o.y = (v[i].y - mean) * rstd * gm[i].y;
        o.z = (v[i].z - mean) * rstd * gm[i].z;
        o.w = (v[i].w - mean) * rstd * gm[i].w;
        if (beta) { o.x += bt[i].x; o.y += bt[i].y; o.z += bt[i].z; o.w += bt[i].w; }
        if (y) {
          uint2 pk = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
          *reinterpret_cast<uint2*>(y + (size_t)row * ldy + 4 * c4) = pk;
        }
        if (yf) *reinterpret_cast<float4*>(yf + (size_t)row * ldyf + 4 * c4) = o;
      }
    }
    if (more) {
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] = nx[i];
    }
  }
}

__device__ __forceinline__ float4 load_dy4(const bf16_t* dyb, const float* dyf, size_t off) {
  if (dyf) return *reinterpret_cast<const float4*>(dyf + off);
  return load_bf16x4(dyb + off);
}

// grid = nblk blocks of 4 waves; wave w of block b walks rows b*4+w, +4*nblk, ...
// partial dgamma/dbeta per block -> ws[b][0][C], ws[b][1][C]
template <int NV>
__global__ __launch_bounds__(256, (NV <= 3 ? LN_BWD_OCC : NV == 4 ? 3 : NV <= 6 ? 2 : 1)) void ln_bwd_kernel(const bf16_t* __restrict__ dyb, int lddy,
                                                      const float* __restrict__ dyf, int lddyf,
                                                      const float* __restrict__ x, const bf16_t* __restrict__ xb, int ldx,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ mean,
                                                      const float* __restrict__ rstd,
                                                      const float* __restrict__ dres, const bf16_t* __restrict__ dresb, int lddres,
                                                      float* dxf, int lddx, bf16_t* dxb, int lddxb,
                                                      float* ws, int M, int C) {
  __shared__ float red[4 * 2048];  // [wave][C <= 2048], reused for dgamma then dbeta
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C4 = C >> 2;
  float4 gm[NV], dg[NV], db[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = lane + 64 * i;
    gm[i] = (c4 < C4) ? reinterpret_cast<const float4*>(gamma)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int ldy_ = dyf ? lddyf : lddy;
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    float4 xh[NV], dy[NV], rv[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {  // the residual-path gradient is fetched with the other operands, not after the reduction
      const int c4 = lane + 64 * i;
      if (dresb) rv[i] = (c4 < C4) ? load_bf16x4(dresb + (size_t)row * lddres + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
      else rv[i] = (dres && c4 < C4) ? *reinterpret_cast<const float4*>(dres + (size_t)row * lddres + 4 * c4)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < C4) {
        float4 xv;
        if (xb) {
          xv = load_bf16x4(xb + (size_t)row * ldx + 4 * c4);
        } else {
          const f32x4_t xv_ = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(x + (size_t)row * ldx + 4 * c4));
          xv = make_float4(xv_[0], xv_[1], xv_[2], xv_[3]);
        }
        dy[i] = load_dy4(dyb, dyf, (size_t)row * ldy_ + 4 * c4);
        xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
        const float a = dy[i].x * gm[i].x, b = dy[i].y * gm[i].y, c = dy[i].z * gm[i].z, d = dy[i].w * gm[i].w;
        s1 += (a + b) + (c + d);
        s2 += (a * xh[i].x + b * xh[i].y) + (c * xh[i].z + d * xh[i].w);
        dg[i].x += dy[i].x * xh[i].x; dg[i].y += dy[i].y * xh[i].y;
        dg[i].z += dy[i].z * xh[i].z; dg[i].w += dy[i].w * xh[i].w;
        db[i].x += dy[i].x; db[i].y += dy[i].y; db[i].z += dy[i].z; db[i].w += dy[i].w;
      }
    }
    const float c1 = wave_sum(s1) / (float)C, c2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < C4) {
        float4 o;
        o.x = rs * (dy[i].x * gm[i].x - c1 - xh[i].x * c2);
        o.y = rs * (dy[i].y * gm[i].y - c1 - xh[i].y * c2);
        o.z = rs * (dy[i].z * gm[i].z - c1 - xh[i].z * c2);
        o.w = rs * (dy[i].w * gm[i].w - c1 - xh[i].w * c2);
        o.x += rv[i].x; o.y += rv[i].y; o.z += rv[i].z; o.w += rv[i].w;
        if (dxf) __builtin_nontemporal_store(f32x4_t{o.x, o.y, o.z, o.w}, reinterpret_cast<f32x4_t*>(dxf + (size_t)row * lddx + 4 * c4));
        if (dxb) {
          uint2 pk = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
          __builtin_nontemporal_store(i32x2_t{(int)pk.x, (int)pk.y}, reinterpret_cast<i32x2_t*>(dxb + (size_t)row * lddxb + 4 * c4));
        }
      }
    }
  }
  // combine the 4 waves' column partials through LDS in two rounds (dgamma, then dbeta) to stay
  // inside 4 x 2048 floats = 32 KiB: red viewed as [4][2048]
  float* r = red;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < C4) *reinterpret_cast<float4*>(r + wave * 2048 + 4 * c4) = pass ? db[i] : dg[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      const float t = (r[c] + r[2048 + c]) + (r[4096 + c] + r[6144 + c]);
      ws[((size_t)blockIdx.x * 2 + pass) * C + c] = t;
    }
  }
}

// LEAN form for the towers' backward (round 5): dy, dres and the output are bf16 (the bf16 gradient stream), x fp32 or bf16.  Same
// formulas per element as ln_bwd_kernel (outputs equal up to hipcc's choice of fused multiply-adds); what differs is the memory side: a bf16 stream moves 8 bytes per
// lane and instruction, so the generic kernel — one row per wave in flight, 4 waves per SIMD — kept too few bytes in flight once the
// fp32 streams were gone (10 bytes per element at 3.9 TB/s where the 16-byte form ran 5.5).  Here the NEXT row's loads are requested
// before the current row's reductions (raw packed words: 8 registers per float4 group) and the kernel is compiled for 3 waves per SIMD.
template <int NV, bool XB>
__global__ __launch_bounds__(256, (NV <= 3 ? 3 : NV == 4 ? 2 : 1)) void ln_bwd_lean_kernel(const bf16_t* __restrict__ dyb, int lddy,
                                                      const void* __restrict__ xv_, int ldx,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ mean,
                                                      const float* __restrict__ rstd,
                                                      const bf16_t* __restrict__ dresb, int lddres,
                                                      bf16_t* dxb, int lddxb, float* ws, int nparts, int M, int C) {
  __shared__ float red[4 * 2048];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C4 = C >> 2;
  // the workspace holds `nparts` rows of partials (the generic kernel's block count: the callers size and reduce it by that); this
  // grid is smaller (one resident round at 3 waves per SIMD), so the rows it does not own are written as zeros
  for (int rz = gridDim.x + blockIdx.x; rz < nparts; rz += gridDim.x)
    for (int c = threadIdx.x; c < 2 * C; c += 256) ws[(size_t)rz * 2 * C + c] = 0.f;
  float4 gm[NV], dg[NV], db[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = lane + 64 * i;
    gm[i] = (c4 < C4) ? reinterpret_cast<const float4*>(gamma)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  struct Raw { uint2 dy[NV], rv[NV]; f32x4_t xf[XB ? 1 : NV]; uint2 xb[XB ? NV : 1]; };
  auto load_row = [&](int row, Raw& r) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c4 = lane + 64 * i;
      const bool ok = c4 < C4;
      r.rv[i] = (dresb && ok) ? *reinterpret_cast<const uint2*>(dresb + (size_t)row * lddres + 4 * c4) : make_uint2(0u, 0u);
      r.dy[i] = ok ? *reinterpret_cast<const uint2*>(dyb + (size_t)row * lddy + 4 * c4) : make_uint2(0u, 0u);
      if constexpr (XB) r.xb[i] = ok ? *reinterpret_cast<const uint2*>((const bf16_t*)xv_ + (size_t)row * ldx + 4 * c4) : make_uint2(0u, 0u);
      else r.xf[i] = ok ? __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>((const float*)xv_ + (size_t)row * ldx + 4 * c4))
                        : f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto widen = [](uint2 pk) {
    return make_float4(bf16_to_f32((bf16_t)(pk.x & 0xffff)), bf16_to_f32((bf16_t)(pk.x >> 16)),
                       bf16_to_f32((bf16_t)(pk.y & 0xffff)), bf16_to_f32((bf16_t)(pk.y >> 16)));
  };
  const int stride = gridDim.x * 4;
  int row = blockIdx.x * 4 + wave;
  Raw cur, nxt;
  if (row < M) load_row(row, cur);
  for (; row < M; row += stride) {
    const bool more = row + stride < M;
    if (more) load_row(row + stride, nxt);
    const float mu = mean[row], rs = rstd[row];
    float4 xh[NV], dy[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < C4) {
        float4 xv;
        if constexpr (XB) xv = widen(cur.xb[i]);
        else xv = make_float4(cur.xf[i][0], cur.xf[i][1], cur.xf[i][2], cur.xf[i][3]);
        dy[i] = widen(cur.dy[i]);
        xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
        const float a = dy[i].x * gm[i].x, b = dy[i].y * gm[i].y, c = dy[i].z * gm[i].z, d = dy[i].w * gm[i].w;
        s1 += (a + b) + (c + d);
        s2 += (a * xh[i].x + b * xh[i].y) + (c * xh[i].z + d * xh[i].w);
        dg[i].x += dy[i].x * xh[i].x; dg[i].y += dy[i].y * xh[i].y;
        dg[i].z += dy[i].z * xh[i].z; dg[i].w += dy[i].w * xh[i].w;
        db[i].x += dy[i].x; db[i].y += dy[i].y; db[i].z += dy[i].z; db[i].w += dy[i].w;
      }
    }
    const float c1 = wave_sum(s1) / (float)C, c2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < C4) {
        const float4 rv = widen(cur.rv[i]);
        float4 o;
        o.x = rs * (dy[i].x * gm[i].x - c1 - xh[i].x * c2);
        o.y = rs * (dy[i].y * gm[i].y - c1 - xh[i].y * c2);
        o.z = rs * (dy[i].z * gm[i].z - c1 - xh[i].z * c2);
        o.w = rs * (dy[i].w * gm[i].w - c1 - xh[i].w * c2);
        o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
        uint2 pk = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
        __builtin_nontemporal_store(i32x2_t{(int)pk.x, (int)pk.y}, reinterpret_cast<i32x2_t*>(dxb + (size_t)row * lddxb + 4 * c4));
      }
    }
    if (more) cur = nxt;
  }
  float* r = red;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < C4) *reinterpret_cast<float4*>(r + wave * 2048 + 4 * c4) = pass ? db[i] : dg[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      const float t = (r[c] + r[2048 + c]) + (r[4096 + c] + r[6144 + c]);
      ws[((size_t)blockIdx.x * 2 + pass) * C + c] = t;
    }
  }
}

// out[c] (+)= sum_b ws[b][which][c]; grid over columns
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ ws, int nblk, int C,
                                                             float* dgamma, float* dbeta, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float g = 0.f, b = 0.f;
  for (int k = 0; k < nblk; ++k) {
    g += ws[((size_t)k * 2 + 0) * C + c];
    b += ws[((size_t)k * 2 + 1) * C + c];
  }
  if (dgamma) dgamma[c] = accumulate ? dgamma[c] + g : g;
  if (dbeta) dbeta[c] = accumulate ? dbeta[c] + b : b;
}

// one resident round of 4-wave blocks: 256 CUs x the waves per SIMD the instantiation is compiled for
inline int ln_bwd_blocks(int M, int C) {
  const int nv = (C / 4 + 63) / 64;
  const int cap = 256 * (nv <= 3 ? LN_BWD_OCC : nv == 4 ? 3 : nv <= 6 ? 2 : 1);
  int nb = (M + 3) / 4;
  return nb > cap ? cap : nb;
}

}  // namespace

// rows are walked by at most 256 CUs x 8 blocks of four waves (LC2IS_LN_FWD_BLOCKS overrides the cap: A/B)
static int ln_fwd_blocks(int M) {
  static const int cap = getenv("LC2IS_LN_FWD_BLOCKS") ? atoi(getenv("LC2IS_LN_FWD_BLOCKS")) : 2048;
  const int need = (M + 3) / 4;
  return need < cap ? need : (cap > 0 ? cap : need);
}

extern "C" int lc2is_layernorm_fwd(const void* x, int ldx, int x_is_bf16, const float* gamma, const float* beta,
                                   void* y_bf16, int ldy, float* y_f32, int ldyf, float* mean, float* rstd,
                                   int M, int C, float eps, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !gamma || (!y_bf16 && !y_f32)) return LC2IS_ERR_NULL;
  const float* xf = x_is_bf16 ? nullptr : (const float*)x;
  const bf16_t* xb = x_is_bf16 ? (const bf16_t*)x : nullptr;
  if (M <= 0 || C <= 0 || C % 4 || C > LN_MAXV * 256) return LC2IS_ERR_SHAPE;
  if (ldx < C || ldx % 4 || (y_bf16 && (ldy < C || ldy % 4)) || (y_f32 && (ldyf < C || ldyf % 4)))
    return LC2IS_ERR_SHAPE;
#define LN_FWD(NV_)                                                                                  \
  hipLaunchKernelGGL(ln_fwd_kernel<NV_>, dim3(ln_fwd_blocks(M)), dim3(256), 0, stream, xf, xb, ldx, gamma, beta, \
                     (bf16_t*)y_bf16, ldy, y_f32, ldyf, mean, rstd, M, C, eps)
  const int nv = (C / 4 + 63) / 64;
  if (nv <= 1) LN_FWD(1); else if (nv == 2) LN_FWD(2); else if (nv == 3) LN_FWD(3);
  else if (nv == 4) LN_FWD(4); else if (nv <= 6) LN_FWD(6); else LN_FWD(8);
#undef LN_FWD
  return lc2is_check_launch();
}

extern "C" size_t lc2is_layernorm_bwd_workspace_bytes(int M, int C) {
  if (M <= 0 || C <= 0) return 0;
  return (size_t)ln_bwd_blocks(M, C) * 2 * (size_t)C * sizeof(float);
}

extern "C" int lc2is_layernorm_bwd(const void* dy_bf16, int lddy, const float* dy_f32, int lddyf,
                                   const void* x, int ldx, int x_is_bf16, const float* gamma, const float* mean,
                                   const float* rstd, const void* dres, int lddres, int dres_is_bf16, float* dx_f32,
                                   int lddx, void* dx_bf16, int lddxb, float* dgamma, float* dbeta, int accumulate,
                                   int M, int C, void* workspace, size_t workspace_bytes,
                                   lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const float* xf = x_is_bf16 ? nullptr : (const float*)x;
  const bf16_t* xb = x_is_bf16 ? (const bf16_t*)x : nullptr;
  const float* dresf = dres_is_bf16 ? nullptr : (const float*)dres;
  const bf16_t* dresb = dres_is_bf16 ? (const bf16_t*)dres : nullptr;
  if ((!dy_bf16 && !dy_f32) || !x || !gamma || !mean || !rstd || (!dx_f32 && !dx_bf16)) return LC2IS_ERR_NULL;
  if (M <= 0 || C <= 0 || C % 4 || C > LN_MAXV * 256) return LC2IS_ERR_SHAPE;
  if (ldx < C || ldx % 4) return LC2IS_ERR_SHAPE;
  if ((dy_f32 ? (lddyf < C || lddyf % 4) : (lddy < C || lddy % 4))) return LC2IS_ERR_SHAPE;
  if ((dres && (lddres < C || lddres % 4)) || (dx_f32 && (lddx < C || lddx % 4)) ||
      (dx_bf16 && (lddxb < C || lddxb % 4)))
    return LC2IS_ERR_SHAPE;
  if (!workspace || workspace_bytes < lc2is_layernorm_bwd_workspace_bytes(M, C)) return LC2IS_ERR_WORKSPACE;
  const int nblk = ln_bwd_blocks(M, C);
  const int nv = (C / 4 + 63) / 64;
  // the towers' path: bf16 dy / dres / output only -> the lean kernel with next-row prefetch (LC2IS_LN_BWD_LEAN=0: the generic one)
  static const bool lean_on = !(getenv("LC2IS_LN_BWD_LEAN") && atoi(getenv("LC2IS_LN_BWD_LEAN")) == 0);
  if (lean_on && dy_bf16 && !dy_f32 && !dx_f32 && dx_bf16 && (!dres || dres_is_bf16) && nv <= 4) {
    const int cap = 256 * (nv <= 2 ? 4 : nv == 3 ? 3 : 2);     // one resident round of 4-wave blocks at the lean kernel's occupancy (116 / 164 / 204 registers)
    const int lgrid = nblk < cap ? nblk : cap;
#define LN_LEAN(NV_)                                                                                                       \
  do {                                                                                                                     \
    if (x_is_bf16)                                                                                                         \
      hipLaunchKernelGGL((ln_bwd_lean_kernel<NV_, true>), dim3(lgrid), dim3(256), 0, stream, (const bf16_t*)dy_bf16, lddy, x, \
                         ldx, gamma, mean, rstd, (const bf16_t*)dres, lddres, (bf16_t*)dx_bf16, lddxb, (float*)workspace, nblk, M, C); \
    else                                                                                                                   \
      hipLaunchKernelGGL((ln_bwd_lean_kernel<NV_, false>), dim3(lgrid), dim3(256), 0, stream, (const bf16_t*)dy_bf16, lddy, x, \
                         ldx, gamma, mean, rstd, (const bf16_t*)dres, lddres, (bf16_t*)dx_bf16, lddxb, (float*)workspace, nblk, M, C); \
  } while (0)
    if (nv <= 1) LN_LEAN(1); else if (nv == 2) LN_LEAN(2); else if (nv == 3) LN_LEAN(3); else LN_LEAN(4);
#undef LN_LEAN
    int rc = lc2is_check_launch();
    if (rc) return rc;
    if (dgamma || dbeta) {
      hipLaunchKernelGGL(partials_reduce_kernel, dim3((C + 31) / 32, 2), dim3(1024), 0, stream,
                         (const float*)workspace, nblk, (size_t)2 * C, (size_t)C, C, dgamma, dbeta, accumulate);
      rc = lc2is_check_launch();
    }
    return rc;
  }
#define LN_BWD(NV_)                                                                                   \
  hipLaunchKernelGGL(ln_bwd_kernel<NV_>, dim3(nblk), dim3(256), 0, stream, (const bf16_t*)dy_bf16, lddy, \
                     dy_f32, lddyf, xf, xb, ldx, gamma, mean, rstd, dresf, dresb, lddres, dx_f32, lddx,  \
                     (bf16_t*)dx_bf16, lddxb, (float*)workspace, M, C)
  if (nv <= 1) LN_BWD(1); else if (nv == 2) LN_BWD(2); else if (nv == 3) LN_BWD(3);
  else if (nv == 4) LN_BWD(4); else if (nv <= 6) LN_BWD(6); else LN_BWD(8);
#undef LN_BWD
  int rc = lc2is_check_launch();
  if (rc) return rc;
  if (dgamma || dbeta) {
    hipLaunchKernelGGL(partials_reduce_kernel, dim3((C + 31) / 32, 2), dim3(1024), 0, stream,
                       (const float*)workspace, nblk, (size_t)2 * C, (size_t)C, C, dgamma, dbeta, accumulate);
    rc = lc2is_check_launch();
  }
  return rc;
}

// The per-block partial sums a lc2is_layernorm_bwd call with dgamma == dbeta == NULL leaves in its workspace:
// `lc2is_layernorm_bwd_partials(M, C)` rows of [dgamma partial (C) | dbeta partial (C)].
extern "C" int lc2is_layernorm_bwd_partials(int M, int C) {
  if (M <= 0 || C <= 0 || C % 4 || C > LN_MAXV * 256) return 0;
  return ln_bwd_blocks(M, C);
}

namespace {
struct LnPartialsGroup {
  lc2is_ln_partials item[LC2IS_LN_PARTIALS_MAX];
};
__global__ __launch_bounds__(1024) void ln_partials_reduce_grouped_kernel(LnPartialsGroup g) {
  const lc2is_ln_partials& it = g.item[blockIdx.z];
  if ((int)blockIdx.x * 32 >= it.C) return;
  float* out = blockIdx.y ? it.dbeta : it.dgamma;
  if (!out) return;
  partials_reduce_body(it.partials + (size_t)blockIdx.y * it.C, it.nparts, (size_t)2 * it.C, it.C, out, it.accumulate,
                       blockIdx.x);
}
}  // namespace

// The dgamma / dbeta reductions of up to LC2IS_LN_PARTIALS_MAX LayerNorm backward calls in ONE launch (descriptors travel
// as kernel arguments; fixed-order sums, so the result is the one the per-call reduce gives).  Two items must not share
// an output vector.
extern "C" int lc2is_ln_partials_reduce(const lc2is_ln_partials* items, int n, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!items) return LC2IS_ERR_NULL;
  if (n <= 0 || n > LC2IS_LN_PARTIALS_MAX) return LC2IS_ERR_SHAPE;
  LnPartialsGroup g{};
  int cmax = 0;
  for (int i = 0; i < n; ++i) {
    const lc2is_ln_partials& it = items[i];
    if (!it.partials || (!it.dgamma && !it.dbeta)) return LC2IS_ERR_NULL;
    if (it.nparts <= 0 || it.C <= 0 || it.C % 4) return LC2IS_ERR_SHAPE;
    for (int j = 0; j < i; ++j)
      if ((it.dgamma && (it.dgamma == items[j].dgamma || it.dgamma == items[j].dbeta)) ||
          (it.dbeta && (it.dbeta == items[j].dgamma || it.dbeta == items[j].dbeta)))
        return LC2IS_ERR_UNSUPPORTED;
    g.item[i] = it;
    cmax = it.C > cmax ? it.C : cmax;
  }
  hipLaunchKernelGGL(ln_partials_reduce_grouped_kernel, dim3((cmax + 31) / 32, 2, n), dim3(1024), 0, stream, g);
  return lc2is_check_launch();
}
