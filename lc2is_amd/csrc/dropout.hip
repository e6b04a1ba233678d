// Dropout / drop-path of the hot path's training mode, without a stored mask (counter-based RNG, common.h).
// replaces: the nn.Dropout modules of torch TransformerDecoderLayer / TransformerEncoderLayer (dropout, dropout1-3;
//   torch:nn/modules/transformer.py:1158-1199, :960-975) as instantiated by PromptLayer (model/decoder.py:26), the SR layers
//   (model/hierarchical.py:176,203, model/decoder.py:115) and ftn.Transformer (model/ftn.py:135), and hf SwinDropPath
//   (modeling_swin.py:280-302, per-SAMPLE Bernoulli) behind SwinTransformer (model/encoder.py:126-127, drop_path_rate 0.1).
// Every kernel is one coalesced pass (16 B per lane), HBM-bound.
#include "common.h"
#include "lc2is_hip.h"

namespace {

inline int dr_grid(size_t items) {
  size_t b = (items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 65535 * 16 ? 65535 * 16 : b));
}

// y = resid + keep * x / (1 - p)       (fp32 x; outputs: fp32 and / or bf16).  rows_per_sample > 0: ONE decision per sample
// (drop-path: row r belongs to sample r / rows_per_sample, coordinate (sample, 0)); otherwise one decision per element (r, c).
__global__ __launch_bounds__(256) void dropout_rows_f32_kernel(const float* __restrict__ x, int ldx,
                                                                const float* __restrict__ resid, int ldr, float* y32,
                                                                int ldy, bf16_t* y16, int ldy16, int M, int C,
                                                                int rows_per_sample, DropCfg d) {
  const int C4 = C >> 2;
  const size_t total = (size_t)M * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int m = (int)(i / C4), c = 4 * (int)(i % C4);
    const f32x4_t v = *(const f32x4_t*)(x + (size_t)m * ldx + c);
    f32x4_t o;
    if (rows_per_sample > 0) {
      const bool k = drop_keep(d, drop_row_hash(d, (unsigned)(m / rows_per_sample)), 0u);
      const float sc = k ? d.inv_keep : 0.f;
      o = v * sc;
    } else {
      const unsigned rh = drop_row_hash(d, (unsigned)m);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = drop_keep(d, rh, (unsigned)(c + e)) ? v[e] * d.inv_keep : 0.f;
    }
    if (resid) o += *(const f32x4_t*)(resid + (size_t)m * ldr + c);
    if (y32) *(f32x4_t*)(y32 + (size_t)m * ldy + c) = o;
    if (y16) *(uint2*)(y16 + (size_t)m * ldy16 + c) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
  }
}

// bf16 in -> bf16 out (in place allowed): the feed-forward hidden dropout and its backward.
__global__ __launch_bounds__(256) void dropout_rows_bf16_kernel(const bf16_t* __restrict__ x, int ldx, bf16_t* y, int ldy,
                                                                 int M, int C, DropCfg d) {
  const int C4 = C >> 2;
  const size_t total = (size_t)M * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int m = (int)(i / C4), c = 4 * (int)(i % C4);
    const uint2 v = *(const uint2*)(x + (size_t)m * ldx + c);
    const unsigned rh = drop_row_hash(d, (unsigned)m);
    float f[4] = {bf16_to_f32((bf16_t)(v.x & 0xffff)), bf16_to_f32((bf16_t)(v.x >> 16)),
                  bf16_to_f32((bf16_t)(v.y & 0xffff)), bf16_to_f32((bf16_t)(v.y >> 16))};
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = drop_keep(d, rh, (unsigned)(c + e)) ? f[e] * d.inv_keep : 0.f;
    *(uint2*)(y + (size_t)m * ldy + c) = make_uint2(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]));
  }
}

// the mask itself, for tests and debugging only (the product never stores one): out[r][c] = keep(seed, r, c) as 0 / 1
__global__ __launch_bounds__(256) void dropout_mask_kernel(unsigned char* out, size_t rows, int cols, DropCfg d) {
  const size_t total = rows * (size_t)cols;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const unsigned r = (unsigned)(i / cols), c = (unsigned)(i % cols);
    out[i] = drop_keep(d, drop_row_hash(d, r), c) ? 1 : 0;
  }
}

}  // namespace

extern "C" int lc2is_dropout_rows_f32(const float* x, int ldx, const float* resid, int ldr, float* y32, int ldy,
                                      void* y16, int ldy16, int M, int C, int rows_per_sample, float p,
                                      unsigned long long seed, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || (!y32 && !y16)) return LC2IS_ERR_NULL;
  if (M <= 0 || C <= 0 || C % 4 || ldx < C || ldx % 4 || (resid && (ldr < C || ldr % 4)) || (y32 && (ldy < C || ldy % 4)) ||
      (y16 && (ldy16 < C || ldy16 % 4)) || rows_per_sample < 0)
    return LC2IS_ERR_SHAPE;
  if (!(p >= 0.f && p < 1.f)) return LC2IS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(dropout_rows_f32_kernel, dim3(dr_grid((size_t)M * C / 4)), dim3(256), 0, stream, x, ldx, resid, ldr, y32,
                     ldy, (bf16_t*)y16, ldy16, M, C, rows_per_sample, make_drop_cfg(p, seed));
  return lc2is_check_launch();
}

extern "C" int lc2is_dropout_rows_bf16(const void* x, int ldx, void* y, int ldy, int M, int C, float p,
                                       unsigned long long seed, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !y) return LC2IS_ERR_NULL;
  if (M <= 0 || C <= 0 || C % 4 || ldx < C || ldx % 4 || ldy < C || ldy % 4) return LC2IS_ERR_SHAPE;
  if (!(p >= 0.f && p < 1.f)) return LC2IS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(dropout_rows_bf16_kernel, dim3(dr_grid((size_t)M * C / 4)), dim3(256), 0, stream, (const bf16_t*)x, ldx,
                     (bf16_t*)y, ldy, M, C, make_drop_cfg(p, seed));
  return lc2is_check_launch();
}

extern "C" int lc2is_dropout_mask(unsigned char* out, long rows, int cols, float p, unsigned long long seed,
                                  lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!out) return LC2IS_ERR_NULL;
  if (rows <= 0 || cols <= 0 || rows > 0xffffffffL) return LC2IS_ERR_SHAPE;
  if (!(p >= 0.f && p < 1.f)) return LC2IS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(dr_grid((size_t)rows * cols)), dim3(256), 0, stream, out, (size_t)rows, cols,
                     make_drop_cfg(p, seed));
  return lc2is_check_launch();
}
