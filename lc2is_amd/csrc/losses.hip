// Remaining per-pixel losses of model/loss.py and the parity metric of metrics.py, as single-pass HIP kernels
// over channels-last [B, HW, K] class scores (gfx950; HBM-bound, fp32 math).
//   * ContrastiveLoss (model/loss.py:39-64): loss_visual = softmax-CE over the K classes of every pixel;
//     loss_textual = nn.CrossEntropyLoss on the [B,H,W,K] view with one-hot FLOAT targets — torch takes dim 1
//     (=H!) as the class axis there, so it is a log-softmax over the image rows per (b, w, k) column.
//   * NPairLoss (model/loss.py:23-37): pos/(pos + sum(neg)) of raw dot products.
//   * compute_mIOU (metrics.py:82-102): argmax of the x4 upsampled scores vs nearest-x4 labels -> per-image
//     intersection / prediction / label counts per class (the IoU arithmetic over <=151 classes stays on the host).
#include "common.h"
#include "lc2is_hip.h"

namespace {

// ---- visual part: per row (pixel) CE over K contiguous scores; one wave per row --------------------------
__global__ __launch_bounds__(256) void rows_ce_kernel(const float* __restrict__ x, const int64_t* __restrict__ labels,
                                                       float* lse_out, float* loss_sum, float* dx, float gscale, int M,
                                                       int K, int accumulate_dx) {
  const int lane = threadIdx.x & 63;
  const int nw = gridDim.x * 4;
  float lacc = 0.f;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += nw) {
    const float* xr = x + (size_t)row * K;
    float m = -__builtin_inff();
    for (int c = lane; c < K; c += 64) m = fmaxf(m, xr[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < K; c += 64) s += __expf(xr[c] - m);
    s = wave_sum(s);
    const float lse = m + __logf(s);
    const int lab = (int)labels[row];
    if (lse_out && lane == 0) lse_out[row] = lse;
    lacc += lse - xr[lab];
    if (dx) {
      float* dr = dx + (size_t)row * K;
      for (int c = lane; c < K; c += 64) {
        const float g = gscale * (__expf(xr[c] - lse) - (c == lab ? 1.f : 0.f));
        dr[c] = accumulate_dx ? dr[c] + g : g;
      }
    }
  }
  if (loss_sum && lane == 0 && lacc != 0.f) atomicAdd(loss_sum, lacc);  // one atomic per wave
}

// ---- textual part: log-softmax over H per (b, w, k) column; thread per column, coalesced over k ----------
__global__ __launch_bounds__(256) void cols_ce_kernel(const float* __restrict__ x, const int64_t* __restrict__ labels,
                                                       float* loss_sum, float* dx, float gscale, int B, int H, int W,
                                                       int K) {
  const size_t total = (size_t)B * W * K;
  float lacc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i % K);
    const int w = (int)((i / K) % W);
    const int b = (int)(i / ((size_t)K * W));
    const float* col = x + (((size_t)b * H) * W + w) * K + k;       // stride over h: W*K
    const int64_t* lcol = labels + ((size_t)b * H) * W + w;          // stride over h: W
    const size_t sx = (size_t)W * K;
    float m = -__builtin_inff();
    for (int h = 0; h < H; ++h) m = fmaxf(m, col[h * sx]);
    float s = 0.f;
    for (int h = 0; h < H; ++h) s += __expf(col[h * sx] - m);
    const float lse = m + __logf(s);
    int cnt = 0;
    for (int h = 0; h < H; ++h) {
      if ((int)lcol[(size_t)h * W] == k) {
        lacc += lse - col[h * sx];
        ++cnt;
      }
    }
    if (dx) {
      float* dcol = dx + (((size_t)b * H) * W + w) * K + k;
      for (int h = 0; h < H; ++h) {
        const float p = __expf(col[h * sx] - lse);
        dcol[h * sx] += gscale * (p * (float)cnt - ((int)lcol[(size_t)h * W] == k ? 1.f : 0.f));
      }
    }
  }
  lacc = wave_sum(lacc);
  if ((threadIdx.x & 63) == 0 && lacc != 0.f) atomicAdd(loss_sum, lacc);
}

// ---- NPairLoss: res[i] = sum_p pos[i][p] / (pos[i][p] + negsum[i]) ; tiny, one wave per row of x ------------
__global__ __launch_bounds__(256) void npair_kernel(const float* __restrict__ x, const float* __restrict__ xp,
                                                     const float* __restrict__ xn, float* res, int n, int np, int nn, int d) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float* xr = x + (size_t)row * d;
  float negsum = 0.f;
  for (int j = 0; j < nn; ++j) {
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += xr[c] * xn[(size_t)j * d + c];
    negsum += wave_sum(s);
  }
  float acc = 0.f;
  for (int j = 0; j < np; ++j) {
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += xr[c] * xp[(size_t)j * d + c];
    const float pos = wave_sum(s);
    acc += pos / (pos + negsum);
  }
  if (lane == 0) res[row] = acc;
}

// ---- NPairLoss backward.  res[i] = sum_p pos_ip / (pos_ip + N_i), pos_ip = x_i . xp_p, N_i = x_i . s, s = sum_q xn_q:
//   d res_i / d pos_ip = N_i / (pos_ip + N_i)^2 =: a_ip,   d res_i / d N_i = -sum_p pos_ip / (pos_ip + N_i)^2 =: b_i.
// Kernel 1 (wave per row i): dx_i = g_i (sum_p a_ip xp_p + b_i s), and the coefficients A[i][p] = g_i a_ip, Bv[i] = g_i b_i.
// Kernel 2 (wave per output row, fixed summation order): dxp_p = sum_i A[i][p] x_i;  dxn_q = sum_i Bv[i] x_i for every q.
__global__ __launch_bounds__(256) void npair_bwd_rows_kernel(const float* __restrict__ x, const float* __restrict__ xp,
                                                              const float* __restrict__ xn, const float* __restrict__ g,
                                                              float* dx, float* A, float* Bv, int n, int np, int nn, int d) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float* xr = x + (size_t)row * d;
  const float gi = g[row];
  float negsum = 0.f;
  for (int j = 0; j < nn; ++j) {
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += xr[c] * xn[(size_t)j * d + c];
    negsum += wave_sum(s);
  }
  float b = 0.f;
  for (int c = lane; c < d; c += 64) dx[(size_t)row * d + c] = 0.f;
  for (int j = 0; j < np; ++j) {
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += xr[c] * xp[(size_t)j * d + c];
    const float pos = wave_sum(s);
    const float den = pos + negsum;
    const float a = gi * negsum / (den * den);
    b -= gi * pos / (den * den);
    if (lane == 0) A[(size_t)row * np + j] = a;
    for (int c = lane; c < d; c += 64) dx[(size_t)row * d + c] += a * xp[(size_t)j * d + c];
  }
  if (lane == 0) Bv[row] = b;
  for (int c = lane; c < d; c += 64) {
    float sn = 0.f;
    for (int j = 0; j < nn; ++j) sn += xn[(size_t)j * d + c];
    dx[(size_t)row * d + c] += b * sn;
  }
}

__global__ __launch_bounds__(256) void npair_bwd_cols_kernel(const float* __restrict__ x, const float* __restrict__ A,
                                                              const float* __restrict__ Bv, float* dxp, float* dxn, int n,
                                                              int np, int nn, int d) {
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);     // output row: [0, np) -> dxp, np -> the shared dxn row
  if (o > np) return;
  for (int c = lane; c < d; c += 64) {
    float acc = 0.f;
    for (int i = 0; i < n; ++i) acc += (o < np ? A[(size_t)i * np + o] : Bv[i]) * x[(size_t)i * d + c];
    if (o < np) {
      dxp[(size_t)o * d + c] = acc;
    } else {
      for (int q = 0; q < nn; ++q) dxn[(size_t)q * d + c] = acc;
    }
  }
}

// ---- mIoU counts: pred = argmax_k scores_hi[b,k,Y,X], label = labels_lo[b, Y/S, X/S] ------------------------
__global__ __launch_bounds__(256) void miou_counts_kernel(const float* __restrict__ hi, const int64_t* __restrict__ labels,
                                                           int* counts, int B, int K, int H, int W, int S) {
  const size_t total = (size_t)B * H * W;
  const size_t plane = (size_t)H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int X = (int)(i % W), Y = (int)((i / W) % H), b = (int)(i / plane);
    const float* base = hi + (size_t)b * K * plane + (size_t)Y * W + X;
    float best = base[0];
    int arg = 0;
    for (int k = 1; k < K; ++k) {
      const float v = base[(size_t)k * plane];
      if (v > best) { best = v; arg = k; }
    }
    const int lab = (int)labels[((size_t)b * (H / S) + Y / S) * (W / S) + X / S];
    int* c = counts + (size_t)b * 3 * K;
    atomicAdd(c + K + arg, 1);                         // prediction count
    if (lab >= 0 && lab < K) {
      atomicAdd(c + 2 * K + lab, 1);                   // label count
      if (lab == arg) atomicAdd(c + arg, 1);           // intersection
    }
  }
}

inline int ls_grid(size_t items) {
  size_t g = (items + 255) / 256;
  if (g > 8192) g = 8192;
  return g < 1 ? 1 : (int)g;
}

}  // namespace

extern "C" int lc2is_rows_ce(const float* x, const int64_t* labels, float* lse, float* loss_sum, float* dx,
                             float grad_scale, int M, int K, int accumulate_dx, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !labels || (!loss_sum && !dx && !lse)) return LC2IS_ERR_NULL;
  if (M <= 0 || K <= 0) return LC2IS_ERR_SHAPE;
  int grid = (M + 3) / 4;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(rows_ce_kernel, dim3(grid), dim3(256), 0, stream, x, labels, lse, loss_sum, dx, grad_scale, M, K,
                     accumulate_dx);
  return lc2is_check_launch();
}

extern "C" int lc2is_cols_ce(const float* x, const int64_t* labels, float* loss_sum, float* dx, float grad_scale, int B,
                             int H, int W, int K, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !labels || !loss_sum) return LC2IS_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || K <= 0) return LC2IS_ERR_SHAPE;
  hipLaunchKernelGGL(cols_ce_kernel, dim3(ls_grid((size_t)B * W * K)), dim3(256), 0, stream, x, labels, loss_sum, dx,
                     grad_scale, B, H, W, K);
  return lc2is_check_launch();
}

extern "C" int lc2is_npair(const float* x, const float* x_pos, const float* x_neg, float* res, int n, int n_pos,
                           int n_neg, int d, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !x_pos || !x_neg || !res) return LC2IS_ERR_NULL;
  if (n <= 0 || n_pos <= 0 || n_neg <= 0 || d <= 0) return LC2IS_ERR_SHAPE;
  hipLaunchKernelGGL(npair_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, x, x_pos, x_neg, res, n, n_pos, n_neg, d);
  return lc2is_check_launch();
}

extern "C" int lc2is_npair_bwd(const float* x, const float* x_pos, const float* x_neg, const float* dres, float* dx,
                               float* dx_pos, float* dx_neg, float* workspace, int n, int n_pos, int n_neg, int d,
                               lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !x_pos || !x_neg || !dres || !dx || !dx_pos || !dx_neg || !workspace) return LC2IS_ERR_NULL;
  if (n <= 0 || n_pos <= 0 || n_neg <= 0 || d <= 0) return LC2IS_ERR_SHAPE;
  float* A = workspace;                    // [n, n_pos]
  float* Bv = workspace + (size_t)n * n_pos;  // [n]
  hipLaunchKernelGGL(npair_bwd_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, x, x_pos, x_neg, dres, dx, A, Bv, n,
                     n_pos, n_neg, d);
  int rc = lc2is_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(npair_bwd_cols_kernel, dim3((n_pos + 1 + 3) / 4), dim3(256), 0, stream, x, (const float*)A,
                     (const float*)Bv, dx_pos, dx_neg, n, n_pos, n_neg, d);
  return lc2is_check_launch();
}

extern "C" int lc2is_miou_counts(const float* scores_hi, const int64_t* labels_lo, int* counts, int B, int K, int H,
                                 int W, int S, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!scores_hi || !labels_lo || !counts) return LC2IS_ERR_NULL;
  if (B <= 0 || K <= 0 || H <= 0 || W <= 0 || S <= 0 || H % S || W % S) return LC2IS_ERR_SHAPE;
  hipLaunchKernelGGL(miou_counts_kernel, dim3(ls_grid((size_t)B * H * W)), dim3(256), 0, stream, scores_hi, labels_lo,
                     counts, B, K, H, W, S);
  return lc2is_check_launch();
}
