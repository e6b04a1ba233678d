// Spatial glue of the multi-scale (hierarchical / FTN) decoder — BASELINE config 5 (gfx950).
// All kernels are single coalesced passes over channels-last token tensors [B, h*w, C] (16 B per lane):
//   * bilinear xS upsample forward / backward     (F.interpolate(mode="bilinear", scale_factor=S) between the
//     rearranges of model/hierarchical.py:103-109,146-149,166-170 and model/decoder.py:66-72,106-109)
//   * spatial-reduction gather / scatter           (the im2col of Conv2d(d, d, kernel=2, stride=2),
//     model/hierarchical.py:191,214 — stride == kernel, so it is a pure row permutation feeding an MFMA GEMM)
//   * L2 normalisation over channels fwd / bwd     (F.normalize(dim=C) of model/final.py:353-354 & friends)
//   * n-ary add                                     (torch.stack(...).sum(0), model/hierarchical.py:128-129)
#include "common.h"
#include "lc2is_hip.h"

namespace {

__device__ __forceinline__ void bilin_taps(int dst, float inv_scale, int n_in, int& i0, int& i1, float& w0, float& w1) {
  float src = inv_scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  i1 = i0 < n_in - 1 ? i0 + 1 : i0;
  w1 = src - (float)i0;
  w0 = 1.f - w1;
}

// out[b, Y, X, :] = bilinear(in[b, :, :, :]) ; fp32 in, fp32 and/or bf16 out
__global__ __launch_bounds__(256) void bilinear_up_fwd_kernel(const float* __restrict__ in, float* out, bf16_t* out16,
                                                               int B, int h, int w, int C, int S) {
  const int C4 = C >> 2, H = h * S, W = w * S;
  const float inv = 1.f / (float)S;
  const size_t total = (size_t)B * H * W * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    size_t pix = i / C4;
    const int X = (int)(pix % W), Y = (int)((pix / W) % H), b = (int)(pix / ((size_t)W * H));
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    bilin_taps(Y, inv, h, y0, y1, wy0, wy1);
    bilin_taps(X, inv, w, x0, x1, wx0, wx1);
    const float4* base = reinterpret_cast<const float4*>(in) + (size_t)b * h * w * C4 + c4;
    const float4 a = base[((size_t)y0 * w + x0) * C4], bq = base[((size_t)y0 * w + x1) * C4];
    const float4 c = base[((size_t)y1 * w + x0) * C4], d = base[((size_t)y1 * w + x1) * C4];
    float4 o;
    o.x = wy0 * (wx0 * a.x + wx1 * bq.x) + wy1 * (wx0 * c.x + wx1 * d.x);
    o.y = wy0 * (wx0 * a.y + wx1 * bq.y) + wy1 * (wx0 * c.y + wx1 * d.y);
    o.z = wy0 * (wx0 * a.z + wx1 * bq.z) + wy1 * (wx0 * c.z + wx1 * d.z);
    o.w = wy0 * (wx0 * a.w + wx1 * bq.w) + wy1 * (wx0 * c.w + wx1 * d.w);
    if (out) reinterpret_cast<float4*>(out)[i] = o;
    if (out16) reinterpret_cast<uint2*>(out16)[i] = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
  }
}

// din[b, y, x, :] (+)= sum over the output pixels that tap (y, x)
__global__ __launch_bounds__(256) void bilinear_up_bwd_kernel(const float* __restrict__ dout, float* din, bf16_t* din16,
                                                               int B, int h, int w, int C, int S, int accumulate) {
  const int C4 = C >> 2, H = h * S, W = w * S;
  const float inv = 1.f / (float)S;
  const size_t total = (size_t)B * h * w * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    size_t pix = i / C4;
    const int x = (int)(pix % w), y = (int)((pix / w) % h), b = (int)(pix / ((size_t)w * h));
    const float4* base = reinterpret_cast<const float4*>(dout) + (size_t)b * H * W * C4 + c4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int Y0 = max(0, S * (y - 1)), Y1 = min(H, S * (y + 2));
    const int X0 = max(0, S * (x - 1)), X1 = min(W, S * (x + 2));
    for (int Y = Y0; Y < Y1; ++Y) {
      int a0, a1;
      float u0, u1;
      bilin_taps(Y, inv, h, a0, a1, u0, u1);
      const float wy = (a0 == y ? u0 : 0.f) + (a1 == y ? u1 : 0.f);
      if (wy == 0.f) continue;
      for (int X = X0; X < X1; ++X) {
        int b0, b1;
        float v0, v1;
        bilin_taps(X, inv, w, b0, b1, v0, v1);
        const float wx = (b0 == x ? v0 : 0.f) + (b1 == x ? v1 : 0.f);
        if (wx == 0.f) continue;
        const float4 g = base[((size_t)Y * W + X) * C4];
        const float ww = wy * wx;
        acc.x += ww * g.x; acc.y += ww * g.y; acc.z += ww * g.z; acc.w += ww * g.w;
      }
    }
    if (din) {
      float4* dp = reinterpret_cast<float4*>(din) + i;
      if (accumulate) { const float4 o = *dp; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
      *dp = acc;
    }
    if (din16) reinterpret_cast<uint2*>(din16)[i] = make_uint2(pack_bf16x2(acc.x, acc.y), pack_bf16x2(acc.z, acc.w));
  }
}

// gather: out[(b, y, x)][q*C + c] = in[(b, 2y+i, 2x+j)][c], q = 2i + j      (bf16 rows of C)
// scatter (inverse, same index map): in_grad[(b, 2y+i, 2x+j)][c] = out_grad[(b, y, x)][q*C + c]
template <bool SCATTER>
__global__ __launch_bounds__(256) void sr_gather_kernel(const bf16_t* __restrict__ src, bf16_t* dst, int B, int h, int w,
                                                         int C) {
  const int C8 = C >> 3, h2 = h >> 1, w2 = w >> 1;
  const size_t total = (size_t)B * h2 * w2 * 4 * C8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    size_t r = i / C8;
    const int q = (int)(r & 3);
    r >>= 2;
    const int x = (int)(r % w2), y = (int)((r / w2) % h2), b = (int)(r / ((size_t)w2 * h2));
    const size_t fine = ((size_t)b * h + 2 * y + (q >> 1)) * w + 2 * x + (q & 1);
    const size_t coarse = ((size_t)b * h2 + y) * w2 + x;
    const uint4* s;
    uint4* d;
    if (SCATTER) {
      s = reinterpret_cast<const uint4*>(src + (coarse * 4 + q) * C) + c8;
      d = reinterpret_cast<uint4*>(dst + fine * C) + c8;
    } else {
      s = reinterpret_cast<const uint4*>(src + fine * C) + c8;
      d = reinterpret_cast<uint4*>(dst + (coarse * 4 + q) * C) + c8;
    }
    *d = *s;
  }
}

// dst32[(b, 2y+i, 2x+j)][c] += src16[(b, y, x)][q*C + c]   (backward of the gather, accumulated onto the fp32 stream)
__global__ __launch_bounds__(256) void sr_scatter_add_kernel(const bf16_t* __restrict__ src, float* dst, int B, int h, int w,
                                                              int C) {
  const int C4 = C >> 2, h2 = h >> 1, w2 = w >> 1;
  const size_t total = (size_t)B * h2 * w2 * 4 * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    size_t r = i / C4;
    const int q = (int)(r & 3);
    r >>= 2;
    const int x = (int)(r % w2), y = (int)((r / w2) % h2), b = (int)(r / ((size_t)w2 * h2));
    const size_t fine = ((size_t)b * h + 2 * y + (q >> 1)) * w + 2 * x + (q & 1);
    const size_t coarse = ((size_t)b * h2 + y) * w2 + x;
    const uint2 pk = *(reinterpret_cast<const uint2*>(src + (coarse * 4 + q) * C) + c4);
    float4* d = reinterpret_cast<float4*>(dst + fine * C) + c4;
    float4 v = *d;
    v.x += bf16_to_f32((bf16_t)(pk.x & 0xffff)); v.y += bf16_to_f32((bf16_t)(pk.x >> 16));
    v.z += bf16_to_f32((bf16_t)(pk.y & 0xffff)); v.w += bf16_to_f32((bf16_t)(pk.y >> 16));
    *d = v;
  }
}

// y = x / max(||x||_2, eps) over the last dim; one wave per row; x fp32 -> y fp32 and/or bf16, saves 1/norm
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* y, bf16_t* y16, float* inv_norm,
                                                          int M, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int C4 = C >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * C);
  float s = 0.f;
  for (int c4 = lane; c4 < C4; c4 += 64) {
    const float4 v = xr[c4];
    s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  const float inv = 1.f / fmaxf(sqrtf(wave_sum(s)), eps);
  if (lane == 0 && inv_norm) inv_norm[row] = inv;
  for (int c4 = lane; c4 < C4; c4 += 64) {
    const float4 v = xr[c4];
    const float4 o = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
    if (y) reinterpret_cast<float4*>(y + (size_t)row * C)[c4] = o;
    if (y16) reinterpret_cast<uint2*>(y16 + (size_t)row * C)[c4] = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
  }
}

// dx = inv * (dy - yhat * <dy, yhat>),  yhat = x * inv   (rows clamped by eps have inv = 1/eps: dx = dy/eps)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ inv_norm, float* dx, int M, int C,
                                                          float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int C4 = C >> 2;
  const float inv = inv_norm[row];
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * C);
  const float4* gr = reinterpret_cast<const float4*>(dy + (size_t)row * C);
  float s = 0.f;
  for (int c4 = lane; c4 < C4; c4 += 64) {
    const float4 v = xr[c4], g = gr[c4];
    s += (v.x * g.x + v.y * g.y) + (v.z * g.z + v.w * g.w);
  }
  const bool clamped = inv >= (1.f / eps) * 0.999999f;
  const float tot = wave_sum(s);
  const float dot = clamped ? 0.f : tot * inv * inv;  // <dy, yhat> * inv
  for (int c4 = lane; c4 < C4; c4 += 64) {
    const float4 v = xr[c4], g = gr[c4];
    reinterpret_cast<float4*>(dx + (size_t)row * C)[c4] =
        make_float4(inv * (g.x - v.x * dot), inv * (g.y - v.y * dot), inv * (g.z - v.z * dot), inv * (g.w - v.w * dot));
  }
}

__global__ __launch_bounds__(256) void add_n_kernel(const float* a, const float* b, const float* c, const float* d, float* out,
                                                     bf16_t* out16, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 s = reinterpret_cast<const float4*>(a)[i];
    const float4 v = reinterpret_cast<const float4*>(b)[i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    if (c) { const float4 u = reinterpret_cast<const float4*>(c)[i]; s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w; }
    if (d) { const float4 u = reinterpret_cast<const float4*>(d)[i]; s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w; }
    if (out) reinterpret_cast<float4*>(out)[i] = s;
    if (out16) reinterpret_cast<uint2*>(out16)[i] = make_uint2(pack_bf16x2(s.x, s.y), pack_bf16x2(s.z, s.w));
  }
}

inline int sp_grid(size_t items) {
  size_t g = (items + 255) / 256;
  if (g > 8192) g = 8192;
  return g < 1 ? 1 : (int)g;
}

}  // namespace

extern "C" int lc2is_bilinear_up_fwd(const float* in, float* out_f32, void* out_bf16, int B, int h, int w, int C, int S,
                                     lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!in || (!out_f32 && !out_bf16)) return LC2IS_ERR_NULL;
  if (B <= 0 || h <= 0 || w <= 0 || C <= 0 || C % 4 || S < 1) return LC2IS_ERR_SHAPE;
  hipLaunchKernelGGL(bilinear_up_fwd_kernel, dim3(sp_grid((size_t)B * h * S * w * S * C / 4)), dim3(256), 0, stream, in,
                     out_f32, (bf16_t*)out_bf16, B, h, w, C, S);
  return lc2is_check_launch();
}

extern "C" int lc2is_bilinear_up_bwd(const float* dout, float* din_f32, void* din_bf16, int B, int h, int w, int C, int S,
                                     int accumulate, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!dout || (!din_f32 && !din_bf16)) return LC2IS_ERR_NULL;
  if (B <= 0 || h <= 0 || w <= 0 || C <= 0 || C % 4 || S < 1) return LC2IS_ERR_SHAPE;
  hipLaunchKernelGGL(bilinear_up_bwd_kernel, dim3(sp_grid((size_t)B * h * w * C / 4)), dim3(256), 0, stream, dout, din_f32,
                     (bf16_t*)din_bf16, B, h, w, C, S, accumulate);
  return lc2is_check_launch();
}

extern "C" int lc2is_sr_gather(const void* src_bf16, void* dst_bf16, int B, int h, int w, int C, int scatter,
                               lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!src_bf16 || !dst_bf16) return LC2IS_ERR_NULL;
  if (B <= 0 || h <= 0 || w <= 0 || (h & 1) || (w & 1) || C <= 0 || C % 8) return LC2IS_ERR_SHAPE;
  const int grid = sp_grid((size_t)B * h * w * C / 8);
  if (scatter)
    hipLaunchKernelGGL(sr_gather_kernel<true>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)src_bf16,
                       (bf16_t*)dst_bf16, B, h, w, C);
  else
    hipLaunchKernelGGL(sr_gather_kernel<false>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)src_bf16,
                       (bf16_t*)dst_bf16, B, h, w, C);
  return lc2is_check_launch();
}

extern "C" int lc2is_sr_scatter_add_f32(const void* src_bf16, float* dst_f32, int B, int h, int w, int C,
                                        lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!src_bf16 || !dst_f32) return LC2IS_ERR_NULL;
  if (B <= 0 || h <= 0 || w <= 0 || (h & 1) || (w & 1) || C <= 0 || C % 4) return LC2IS_ERR_SHAPE;
  hipLaunchKernelGGL(sr_scatter_add_kernel, dim3(sp_grid((size_t)B * h * w * C / 4)), dim3(256), 0, stream,
                     (const bf16_t*)src_bf16, dst_f32, B, h, w, C);
  return lc2is_check_launch();
}

extern "C" int lc2is_l2norm_fwd(const float* x, float* y_f32, void* y_bf16, float* inv_norm, int M, int C, float eps,
                                lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || (!y_f32 && !y_bf16)) return LC2IS_ERR_NULL;
  if (M <= 0 || C <= 0 || C % 4) return LC2IS_ERR_SHAPE;
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, x, y_f32, (bf16_t*)y_bf16, inv_norm, M, C,
                     eps);
  return lc2is_check_launch();
}

extern "C" int lc2is_l2norm_bwd(const float* dy, const float* x, const float* inv_norm, float* dx, int M, int C, float eps,
                                lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!dy || !x || !inv_norm || !dx) return LC2IS_ERR_NULL;
  if (M <= 0 || C <= 0 || C % 4) return LC2IS_ERR_SHAPE;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, dy, x, inv_norm, dx, M, C, eps);
  return lc2is_check_launch();
}

extern "C" int lc2is_add_n(const float* a, const float* b, const float* c, const float* d, float* out_f32, void* out_bf16,
                           size_t n, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!a || !b || (!out_f32 && !out_bf16)) return LC2IS_ERR_NULL;
  if (n == 0 || n % 4) return LC2IS_ERR_SHAPE;
  hipLaunchKernelGGL(add_n_kernel, dim3(sp_grid(n / 4)), dim3(256), 0, stream, a, b, c, d, out_f32, (bf16_t*)out_bf16, n / 4);
  return lc2is_check_launch();
}
