// Device-side image / label preprocessing in front of the hot path: what the reference does on the
// host with Pillow inside transformers' CLIPFeatureExtractor (evaluate.py:58-61, data/collator.py:82-91).
// Integer / byte work, HBM-bound, bit-exact against Pillow:
//   * resample_u8: one 8-bit separable pass of PIL's ImagingResample (Resample.c): 22-bit fixed-point coefficients
//     (computed on the host exactly as precompute_coeffs / normalize_coeffs_8bpc do), rounding constant 1 << 21,
//     arithmetic shift, clip to [0,255].  Horizontal then vertical pass through an 8-bit intermediate, like Pillow.
//   * gather2d_u8: nearest-neighbour resize as a gather through host-built index vectors (Geometry.c ImagingScaleAffine).
//   * crop_lut: centre crop + per-channel 256-entry lookup: uint8 HWC -> float32 CHW (rescale + normalise folded into
//     the table with the reference's exact float ops) or uint8 HW -> int64 (label ids).
#include "common.h"
#include "lc2is_hip.h"

namespace {

constexpr int PP_PRECISION_BITS = 32 - 8 - 2;

__global__ __launch_bounds__(256) void resample_u8_kernel(const unsigned char* __restrict__ src, int H, int W, int C,
                                                           unsigned char* dst, int out_size, int axis,
                                                           const int* __restrict__ bounds, const int* __restrict__ kk,
                                                           int ksize) {
  const int oh = axis == 0 ? out_size : H, ow = axis == 1 ? out_size : W;
  const size_t total = (size_t)oh * ow * C;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int c = (int)(idx % C), x = (int)((idx / C) % ow), y = (int)(idx / ((size_t)C * ow));
    const int o = axis == 0 ? y : x;
    const int first = bounds[2 * o], n = bounds[2 * o + 1];
    const int* k = kk + (size_t)o * ksize;
    int acc = 1 << (PP_PRECISION_BITS - 1);
    if (axis == 1) {
      const unsigned char* p = src + ((size_t)y * W + first) * C + c;
      for (int t = 0; t < n; ++t) acc += (int)p[(size_t)t * C] * k[t];
    } else {
      const unsigned char* p = src + ((size_t)first * W + x) * C + c;
      for (int t = 0; t < n; ++t) acc += (int)p[(size_t)t * W * C] * k[t];
    }
    int v = acc >> PP_PRECISION_BITS;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    dst[idx] = (unsigned char)v;
  }
}

__global__ __launch_bounds__(256) void gather2d_u8_kernel(const unsigned char* __restrict__ src, int W, int C,
                                                           unsigned char* dst, int oh, int ow,
                                                           const int* __restrict__ yi, const int* __restrict__ xi) {
  const size_t total = (size_t)oh * ow * C;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int c = (int)(idx % C), x = (int)((idx / C) % ow), y = (int)(idx / ((size_t)C * ow));
    dst[idx] = src[((size_t)yi[y] * W + xi[x]) * C + c];
  }
}

__global__ __launch_bounds__(256) void crop_lut_f32_kernel(const unsigned char* __restrict__ src, int W, int C, int top,
                                                            int left, int S, const float* __restrict__ lut, float* dst) {
  const size_t total = (size_t)C * S * S;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int x = (int)(idx % S), y = (int)((idx / S) % S), c = (int)(idx / ((size_t)S * S));
    dst[idx] = lut[c * 256 + src[((size_t)(top + y) * W + left + x) * C + c]];
  }
}

__global__ __launch_bounds__(256) void crop_lut_i64_kernel(const unsigned char* __restrict__ src, int W, int C, int top,
                                                            int left, int S, const int64_t* __restrict__ lut,
                                                            int64_t* dst) {
  const size_t total = (size_t)S * S;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int x = (int)(idx % S), y = (int)(idx / S);
    dst[idx] = lut[src[((size_t)(top + y) * W + left + x) * C]];   // channel 0 (data/collator.py:91)
  }
}

inline int pp_grid(size_t items) {
  size_t g = (items + 255) / 256;
  if (g > 8192) g = 8192;
  return g < 1 ? 1 : (int)g;
}

}  // namespace

extern "C" int lc2is_resample_u8(const void* src, int H, int W, int C, void* dst, int out_size, int axis,
                                 const int* bounds, const int* kk, int ksize, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!src || !dst || !bounds || !kk) return LC2IS_ERR_NULL;
  if (H <= 0 || W <= 0 || C <= 0 || out_size <= 0 || ksize <= 0 || (axis != 0 && axis != 1)) return LC2IS_ERR_SHAPE;
  const size_t total = (size_t)(axis == 0 ? out_size : H) * (axis == 1 ? out_size : W) * C;
  hipLaunchKernelGGL(resample_u8_kernel, dim3(pp_grid(total)), dim3(256), 0, stream, (const unsigned char*)src, H, W, C,
                     (unsigned char*)dst, out_size, axis, bounds, kk, ksize);
  return lc2is_check_launch();
}

extern "C" int lc2is_gather2d_u8(const void* src, int H, int W, int C, void* dst, int out_h, int out_w, const int* yi,
                                 const int* xi, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!src || !dst || !yi || !xi) return LC2IS_ERR_NULL;
  if (H <= 0 || W <= 0 || C <= 0 || out_h <= 0 || out_w <= 0) return LC2IS_ERR_SHAPE;
  hipLaunchKernelGGL(gather2d_u8_kernel, dim3(pp_grid((size_t)out_h * out_w * C)), dim3(256), 0, stream,
                     (const unsigned char*)src, W, C, (unsigned char*)dst, out_h, out_w, yi, xi);
  return lc2is_check_launch();
}

extern "C" int lc2is_crop_lut(const void* src, int H, int W, int C, int top, int left, int S, const float* lut_f32,
                              float* dst_f32, const int64_t* lut_i64, int64_t* dst_i64, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!src || (!dst_f32 && !dst_i64) || (dst_f32 && !lut_f32) || (dst_i64 && !lut_i64)) return LC2IS_ERR_NULL;
  if (H <= 0 || W <= 0 || C <= 0 || S <= 0 || top < 0 || left < 0 || top + S > H || left + S > W) return LC2IS_ERR_SHAPE;
  if (dst_f32)
    hipLaunchKernelGGL(crop_lut_f32_kernel, dim3(pp_grid((size_t)C * S * S)), dim3(256), 0, stream,
                       (const unsigned char*)src, W, C, top, left, S, lut_f32, dst_f32);
  if (dst_i64)
    hipLaunchKernelGGL(crop_lut_i64_kernel, dim3(pp_grid((size_t)S * S)), dim3(256), 0, stream, (const unsigned char*)src,
                       W, C, top, left, S, lut_i64, dst_i64);
  return lc2is_check_launch();
}
