// Bandwidth-bound glue kernels of the LC2IS hot path (gfx950): every one is a single coalesced pass
// with 16-byte lane accesses; none is shaped into a GEMM.
//   - bf16 shadow refresh of the fp32 master weights (row-major + transposed copy, one launch for ALL weights)
//   - casts / transposes of small activations
//   - ViT patch gather (conv with stride == kernel -> GEMM operand), token assembly, their backward
//   - CLIP text token+position embedding and its backward
//   - per-batch row-range copies (drop / re-insert the CLS token)
//   - fused SGD / AdamW update over the flat parameter arena
#include "common.h"
#include "lc2is_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------------
// shadow refresh: dst[n][k] = bf16(src[n][k]), dstT[k][n] = bf16(src[n][k]) for a table of weights
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void shadow_refresh_kernel(const lc2is_shadow_desc* __restrict__ descs,
                                                              int ndesc) {
  __shared__ bf16_t tile[64][66];
  // find the descriptor owning this block (tile_start is an exclusive prefix sum of tile counts)
  int lo = 0, hi = ndesc - 1;
  const int bid = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].tile_start <= bid) lo = mid; else hi = mid - 1;
  }
  const lc2is_shadow_desc d = descs[lo];
  const int t = bid - d.tile_start;
  const int tk = (d.K + 63) / 64;
  const int n0 = (t / tk) * 64, k0 = (t % tk) * 64;
  const float* src = (const float*)d.src;
  bf16_t* dst = (bf16_t*)d.dst;
  bf16_t* dstT = (bf16_t*)d.dstT;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 16 x 16 threads, each 4 columns x 4 rows
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n0 + ty + 16 * r, k = k0 + 4 * tx;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < d.N && k < d.K) v = *reinterpret_cast<const float4*>(src + (size_t)n * d.K + k);  // K % 4 == 0
    const bf16_t b0 = f32_to_bf16(v.x), b1 = f32_to_bf16(v.y), b2 = f32_to_bf16(v.z), b3 = f32_to_bf16(v.w);
    if (d.flags & 1) {  // plain fp32 copy (fused bias vectors)
      if (dst && n < d.N && k < d.K) *reinterpret_cast<float4*>((float*)d.dst + (size_t)n * d.ld_dst + k) = v;
    } else if (dst && n < d.N && k < d.K) {
      uint2 pk = make_uint2((unsigned)b0 | ((unsigned)b1 << 16), (unsigned)b2 | ((unsigned)b3 << 16));
      *reinterpret_cast<uint2*>(dst + (size_t)n * d.ld_dst + k) = pk;
    }
    tile[ty + 16 * r][4 * tx + 0] = b0;
    tile[ty + 16 * r][4 * tx + 1] = b1;
    tile[ty + 16 * r][4 * tx + 2] = b2;
    tile[ty + 16 * r][4 * tx + 3] = b3;
  }
  if (!dstT) return;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = k0 + ty + 16 * r, n = n0 + 4 * tx;
    if (k < d.K && n < d.N) {  // N % 4 == 0
      const int kk = ty + 16 * r;
      uint2 pk = make_uint2((unsigned)tile[4 * tx][kk] | ((unsigned)tile[4 * tx + 1][kk] << 16),
                            (unsigned)tile[4 * tx + 2][kk] | ((unsigned)tile[4 * tx + 3][kk] << 16));
      *reinterpret_cast<uint2*>(dstT + (size_t)k * d.ld_dstT + n) = pk;
    }
  }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ src, int lds_,
                                                             bf16_t* dst, int ldd, int M, int C) {
  const int C4 = C >> 2;
  const size_t total = (size_t)M * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int m = (int)(i / C4), c4 = (int)(i % C4);
    const float4 v = *reinterpret_cast<const float4*>(src + (size_t)m * lds_ + 4 * c4);
    *reinterpret_cast<uint2*>(dst + (size_t)m * ldd + 4 * c4) =
        make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
  }
}

__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, int lds_,
                                                              bf16_t* dst, int ldd, int R, int C, long bs_src,
                                                              long bs_dst) {
  __shared__ bf16_t tile[64][66];
  src += (size_t)blockIdx.z * bs_src;   // batched form: one matrix per blockIdx.z
  dst += (size_t)blockIdx.z * bs_dst;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int rr = r0 + r, cc = c0 + tx;
    tile[r][tx] = (rr < R && cc < C) ? src[(size_t)rr * lds_ + cc] : (bf16_t)0;
  }
  __syncthreads();
  for (int c = ty; c < 64; c += 4) {
    const int cc = c0 + c, rr = r0 + tx;
    if (cc < C && rr < R) dst[(size_t)cc * ldd + rr] = tile[tx][c];
  }
}

// ------------------------------------------------------------------------------------------------------
// ViT patch gather: out[(b*G*G + gy*G + gx)][c*ps*ps + i*ps + j] = pix[b][c][gy*ps+i][gx*ps+j]
// (k order == flattened conv weight [C_out][3][ps][ps]); columns >= 3*ps*ps are zero padding up to ldo.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ pix, bf16_t* out, int ldo,
                                                        int B, int Himg, int Wimg, int ps, int G) {
  // one block per (b, gy, c, i): a full image row segment of W pixels -> G patches x ps columns
  const int i = blockIdx.x % ps;
  const int c = (blockIdx.x / ps) % 3;
  const int gy = (blockIdx.x / (ps * 3)) % G;
  const int b = blockIdx.x / (ps * 3 * G);
  const float* row = pix + (((size_t)b * 3 + c) * Himg + gy * ps + i) * Wimg;
  for (int x = threadIdx.x; x < G * ps; x += 256) {
    const int gx = x / ps, j = x % ps;
    out[((size_t)b * G * G + gy * G + gx) * ldo + c * ps * ps + i * ps + j] = f32_to_bf16(row[x]);
  }
}

// ps % 8 == 0 and W % 8 == 0 (round 5): a thread takes 8 consecutive pixels of an image row — two 16-byte loads, ONE 16-byte store into
// the patch row they belong to (the scalar form above stores 2 bytes per lane: 58 us for 32 images at 512 x 512 against the ~30 us
// its 150 MB take at the rate the other streaming kernels reach).  Same values: a conversion per pixel.
__global__ __launch_bounds__(256) void patchify8_kernel(const float* __restrict__ pix, bf16_t* out, int ldo,
                                                         int B, int Himg, int Wimg, int ps, int G) {
  const int W8 = Wimg >> 3;
  const size_t total = (size_t)B * 3 * Himg * W8;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
    const int x8 = (int)(t % W8);
    const size_t rowi = t / W8;                    // (b * 3 + c) * Himg + y
    const int y = (int)(rowi % Himg);
    const int c = (int)((rowi / Himg) % 3), b = (int)(rowi / ((size_t)Himg * 3));
    const int gy = y / ps, i = y % ps, x = x8 * 8, gx = x / ps, j = x % ps;
    if (gy >= G || gx >= G) continue;              // (image edge beyond the last whole patch)
    const float4* src = reinterpret_cast<const float4*>(pix + rowi * Wimg + x);
    const float4 a = src[0], d = src[1];
    i32x4_t pk = {(int)pack_bf16x2(a.x, a.y), (int)pack_bf16x2(a.z, a.w), (int)pack_bf16x2(d.x, d.y), (int)pack_bf16x2(d.z, d.w)};
    *reinterpret_cast<i32x4_t*>(out + ((size_t)b * G * G + gy * G + gx) * ldo + c * ps * ps + i * ps + j) = pk;
  }
}

__global__ __launch_bounds__(256) void zero_pad_cols_kernel(bf16_t* out, int ldo, int M, int c_begin) {
  const int w = ldo - c_begin;
  const size_t total = (size_t)M * w;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256)
    out[(i / w) * ldo + c_begin + (i % w)] = 0;
}

// x[b][0] = cls + pos[0]; x[b][1+p] = patch[b*P+p] + pos[1+p]      (fp32 out, patch fp32)
__global__ __launch_bounds__(256) void vit_embed_fwd_kernel(const float* __restrict__ patch, int ldp,
                                                             const float* __restrict__ cls,
                                                             const float* __restrict__ pos, float* x, int ldx,
                                                             int B, int P, int C) {
  const int C4 = C >> 2;
  const size_t total = (size_t)B * (P + 1) * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const int tok = (int)(i / C4);
    const int s = tok % (P + 1), b = tok / (P + 1);
    float4 v = (s == 0) ? reinterpret_cast<const float4*>(cls)[c4]
                        : *reinterpret_cast<const float4*>(patch + ((size_t)b * P + s - 1) * ldp + 4 * c4);
    const float4 pe = *reinterpret_cast<const float4*>(pos + (size_t)s * C + 4 * c4);
    v.x += pe.x; v.y += pe.y; v.z += pe.z; v.w += pe.w;
    *reinterpret_cast<float4*>(x + (size_t)tok * ldx + 4 * c4) = v;
  }
}

// dpos[s] = sum_b dx[b][s]; dcls = sum_b dx[b][0]; dpatch(bf16)[b*P+p] = dx[b][1+p]
__global__ __launch_bounds__(256) void vit_embed_bwd_kernel(const float* __restrict__ dx, int ldx,
                                                             float* dpos, float* dcls, bf16_t* dpatch, int ldp,
                                                             int B, int P, int C, int accumulate) {
  const int C4 = C >> 2;
  const size_t total = (size_t)(P + 1) * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4), s = (int)(i / C4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = 0; b < B; ++b) {
      const float4 v = *reinterpret_cast<const float4*>(dx + ((size_t)b * (P + 1) + s) * ldx + 4 * c4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      if (s > 0 && dpatch)
        *reinterpret_cast<uint2*>(dpatch + ((size_t)b * P + s - 1) * ldp + 4 * c4) =
            make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    }
    float4* dp = reinterpret_cast<float4*>(dpos + (size_t)s * C) + c4;
    if (accumulate) { const float4 o = *dp; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
    *dp = acc;
    if (s == 0) {
      // d cls equals the position-0 column sum of dx (before adding any previous dpos)
      float4 cacc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int b = 0; b < B; ++b) {
        const float4 v = *reinterpret_cast<const float4*>(dx + ((size_t)b * (P + 1)) * ldx + 4 * c4);
        cacc.x += v.x; cacc.y += v.y; cacc.z += v.z; cacc.w += v.w;
      }
      float4* dc = reinterpret_cast<float4*>(dcls) + c4;
      if (accumulate) { const float4 o = *dc; cacc.x += o.x; cacc.y += o.y; cacc.z += o.z; cacc.w += o.w; }
      *dc = cacc;
    }
  }
}

// x[b*L+l] = tok[ids[b][l]] + pos[l]
__global__ __launch_bounds__(256) void text_embed_fwd_kernel(const int64_t* __restrict__ ids,
                                                              const float* __restrict__ tok,
                                                              const float* __restrict__ pos, float* x, int ldx,
                                                              int BL, int L, int C, int vocab) {
  const int C4 = C >> 2;
  const size_t total = (size_t)BL * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4), r = (int)(i / C4);
    int64_t id = ids[r];
    if (id < 0) id = 0;
    if (id >= vocab) id = vocab - 1;
    const float4 t = *reinterpret_cast<const float4*>(tok + (size_t)id * C + 4 * c4);
    const float4 pe = *reinterpret_cast<const float4*>(pos + (size_t)(r % L) * C + 4 * c4);
    *reinterpret_cast<float4*>(x + (size_t)r * ldx + 4 * c4) =
        make_float4(t.x + pe.x, t.y + pe.y, t.z + pe.z, t.w + pe.w);
  }
}

// dtok[id] += sum over the rows r (in ascending order) with ids[r] == id of dx[r]  (dtok holds zeros or the running gradient);
// dpos[l] (+)= sum_b dx[b*L+l].  NO ATOMICS (round 5; rounds 1-4 scattered with fp32 atomics, whose arrival order made the step's
// last bits run-dependent): block r owns token row r.  It first scans ids[0..r) for an earlier occurrence of its id — if there is
// one, the block that owns the FIRST occurrence does the work and this one leaves — then sums the rows r' >= r that carry the id
// in row order and adds the sum to dtok[id] with a plain read-modify-write (the id has exactly one owner).  B*L is a few
// hundred to a few thousand rows: the scans are L2-resident integer reads.
__global__ __launch_bounds__(256) void text_embed_dtok_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dx,
                                                               int ldx, float* dtok, int R, int C, int vocab) {
  const int r = blockIdx.x;
  auto clampid = [&](int64_t v) { return v < 0 ? (int64_t)0 : (v >= vocab ? (int64_t)(vocab - 1) : v); };
  const int64_t id = clampid(ids[r]);
  int seen = 0;
  for (int q = threadIdx.x; q < r; q += 256) seen |= (clampid(ids[q]) == id);
  if (__syncthreads_or(seen)) return;
  for (int c = threadIdx.x; c < C; c += 256) {
    float acc = 0.f;
    for (int q = r; q < R; ++q)                 // (block-uniform condition: one scalar compare per row)
      if (clampid(ids[q]) == id) acc += dx[(size_t)q * ldx + c];
    dtok[(size_t)id * C + c] += acc;
  }
}

__global__ __launch_bounds__(256) void text_embed_bwd_kernel(const float* __restrict__ dx, int ldx, float* dpos, int B, int L,
                                                              int C, int accumulate) {
  const int total = L * C;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int c = i % C, l = i / C;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += dx[((size_t)b * L + l) * ldx + c];
    dpos[(size_t)l * C + c] = accumulate ? dpos[(size_t)l * C + c] + acc : acc;
  }
}

// the same row copy from a bf16 source (the bf16 residual stream of round 5): widened to fp32 and / or copied as bf16
__global__ __launch_bounds__(256) void rows_copy_bf16_kernel(const bf16_t* __restrict__ src, int S_src, int src_off,
                                                              float* dst, bf16_t* dst_b, int S_dst, int dst_off,
                                                              int B, int n, int C) {
  const int C4 = C >> 2;
  const size_t total = (size_t)B * n * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const int r = (int)(i / C4);
    const int s = r % n, b = r / n;
    const uint2 pk = *reinterpret_cast<const uint2*>(src + ((size_t)b * S_src + src_off + s) * C + 4 * c4);
    const size_t o = ((size_t)b * S_dst + dst_off + s) * C + 4 * c4;
    if (dst) *reinterpret_cast<float4*>(dst + o) = make_float4(bf16_to_f32((bf16_t)(pk.x & 0xffff)), bf16_to_f32((bf16_t)(pk.x >> 16)),
                                                               bf16_to_f32((bf16_t)(pk.y & 0xffff)), bf16_to_f32((bf16_t)(pk.y >> 16)));
    if (dst_b) *reinterpret_cast<uint2*>(dst_b + o) = pk;
  }
}

// dst[b][dst_off + s][:] = src[b][src_off + s][:] for s < n; optional bf16 copy; fp32 rows of C floats
__global__ __launch_bounds__(256) void rows_copy_kernel(const float* __restrict__ src, int S_src, int src_off,
                                                         float* dst, bf16_t* dst_b, int S_dst, int dst_off,
                                                         int B, int n, int C) {
  const int C4 = C >> 2;
  const size_t total = (size_t)B * n * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const int r = (int)(i / C4);
    const int s = r % n, b = r / n;
    const float4 v = *reinterpret_cast<const float4*>(src + ((size_t)b * S_src + src_off + s) * C + 4 * c4);
    const size_t o = ((size_t)b * S_dst + dst_off + s) * C + 4 * c4;
    if (dst) *reinterpret_cast<float4*>(dst + o) = v;
    if (dst_b) *reinterpret_cast<uint2*>(dst_b + o) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
  }
}

// ------------------------------------------------------------------------------------------------------
// optimizers over the flat arena
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sgd_kernel(float* p, const float* __restrict__ g, float* mom, size_t n4,
                                                   float lr, float momentum, float wd, float gscale) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float d[4] = {gv.x * gscale + wd * pv.x, gv.y * gscale + wd * pv.y, gv.z * gscale + wd * pv.z,
                  gv.w * gscale + wd * pv.w};
    if (mom) {
      float4 mv = reinterpret_cast<float4*>(mom)[i];
      mv.x = momentum * mv.x + d[0]; mv.y = momentum * mv.y + d[1];
      mv.z = momentum * mv.z + d[2]; mv.w = momentum * mv.w + d[3];
      reinterpret_cast<float4*>(mom)[i] = mv;
      d[0] = mv.x; d[1] = mv.y; d[2] = mv.z; d[3] = mv.w;
    }
    pv.x -= lr * d[0]; pv.y -= lr * d[1]; pv.z -= lr * d[2]; pv.w -= lr * d[3];
    reinterpret_cast<float4*>(p)[i] = pv;
  }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* __restrict__ g, float* m, float* v,
                                                     size_t n4, float lr, float b1, float b2, float eps,
                                                     float wd, float bc1, float bc2, float gscale) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    float* pp = &pv.x; const float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gg = gp[k] * gscale;
      pp[k] *= (1.f - lr * wd);
      mp[k] = b1 * mp[k] + (1.f - b1) * gg;
      vp[k] = b2 * vp[k] + (1.f - b2) * gg * gg;
      const float denom = sqrtf(vp[k]) / sqrtf(bc2) + eps;
      pp[k] -= (lr / bc1) * (mp[k] / denom);
    }
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
}

inline int ew_grid(size_t work_items) {
  size_t g = (work_items + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int lc2is_shadow_refresh(const lc2is_shadow_desc* descs_dev, int ndesc, int total_tiles,
                                    lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!descs_dev) return LC2IS_ERR_NULL;
  if (ndesc <= 0 || total_tiles <= 0) return LC2IS_ERR_SHAPE;
  hipLaunchKernelGGL(shadow_refresh_kernel, dim3(total_tiles), dim3(256), 0, stream, descs_dev, ndesc);
  return lc2is_check_launch();
}

extern "C" int lc2is_cast_f32_bf16(const float* src, int ld_src, void* dst, int ld_dst, int M, int C,
                                   lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!src || !dst) return LC2IS_ERR_NULL;
  if (M <= 0 || C <= 0 || C % 4 || ld_src < C || ld_dst < C || ld_src % 4 || ld_dst % 4) return LC2IS_ERR_SHAPE;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(ew_grid((size_t)M * C / 4)), dim3(256), 0, stream, src, ld_src,
                     (bf16_t*)dst, ld_dst, M, C);
  return lc2is_check_launch();
}

extern "C" int lc2is_transpose_bf16(const void* src, int ld_src, void* dst, int ld_dst, int R, int C,
                                    lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!src || !dst) return LC2IS_ERR_NULL;
  if (R <= 0 || C <= 0 || ld_src < C || ld_dst < R) return LC2IS_ERR_SHAPE;
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, stream,
                     (const bf16_t*)src, ld_src, (bf16_t*)dst, ld_dst, R, C, 0L, 0L);
  return lc2is_check_launch();
}

extern "C" int lc2is_transpose_bf16_batched(const void* src, int ld_src, long stride_src, void* dst, int ld_dst,
                                            long stride_dst, int R, int C, int batch, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!src || !dst) return LC2IS_ERR_NULL;
  if (R <= 0 || C <= 0 || ld_src < C || ld_dst < R || batch <= 0 || batch > 65535) return LC2IS_ERR_SHAPE;
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((C + 63) / 64, (R + 63) / 64, batch), dim3(256), 0, stream,
                     (const bf16_t*)src, ld_src, (bf16_t*)dst, ld_dst, R, C, stride_src, stride_dst);
  return lc2is_check_launch();
}

extern "C" int lc2is_patchify(const float* pixels, void* out_bf16, int ld_out, int B, int H, int W,
                              int patch, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!pixels || !out_bf16) return LC2IS_ERR_NULL;
  if (B <= 0 || H <= 0 || W != H || patch <= 0 || H / patch <= 0) return LC2IS_ERR_SHAPE;
  const int G = H / patch, kdim = 3 * patch * patch;
  if (ld_out < kdim) return LC2IS_ERR_SHAPE;
  if (patch % 8 == 0 && W % 8 == 0 && ld_out % 8 == 0 && ((uintptr_t)pixels & 15) == 0 && ((uintptr_t)out_bf16 & 15) == 0)
    hipLaunchKernelGGL(patchify8_kernel, dim3(ew_grid((size_t)B * 3 * H * W / 8)), dim3(256), 0, stream, pixels, (bf16_t*)out_bf16,
                       ld_out, B, H, W, patch, G);
  else
    hipLaunchKernelGGL(patchify_kernel, dim3(B * G * 3 * patch), dim3(256), 0, stream, pixels, (bf16_t*)out_bf16,
                       ld_out, B, H, W, patch, G);
  int rc = lc2is_check_launch();
  if (rc || ld_out == kdim) return rc;
  hipLaunchKernelGGL(zero_pad_cols_kernel, dim3(ew_grid((size_t)B * G * G * (ld_out - kdim))), dim3(256), 0,
                     stream, (bf16_t*)out_bf16, ld_out, B * G * G, kdim);
  return lc2is_check_launch();
}

extern "C" int lc2is_vit_embed_fwd(const float* patch, int ld_patch, const float* cls, const float* pos,
                                   float* x, int ldx, int B, int P, int C, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!patch || !cls || !pos || !x) return LC2IS_ERR_NULL;
  if (B <= 0 || P <= 0 || C <= 0 || C % 4 || ld_patch < C || ld_patch % 4 || ldx < C || ldx % 4)
    return LC2IS_ERR_SHAPE;
  hipLaunchKernelGGL(vit_embed_fwd_kernel, dim3(ew_grid((size_t)B * (P + 1) * C / 4)), dim3(256), 0, stream,
                     patch, ld_patch, cls, pos, x, ldx, B, P, C);
  return lc2is_check_launch();
}

extern "C" int lc2is_vit_embed_bwd(const float* dx, int ldx, float* dpos, float* dcls, void* dpatch_bf16,
                                   int ld_dpatch, int B, int P, int C, int accumulate, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!dx || !dpos || !dcls) return LC2IS_ERR_NULL;
  if (B <= 0 || P <= 0 || C <= 0 || C % 4 || ldx < C || ldx % 4 || (dpatch_bf16 && (ld_dpatch < C || ld_dpatch % 4)))
    return LC2IS_ERR_SHAPE;
  hipLaunchKernelGGL(vit_embed_bwd_kernel, dim3(ew_grid((size_t)(P + 1) * C / 4)), dim3(256), 0, stream, dx, ldx,
                     dpos, dcls, (bf16_t*)dpatch_bf16, ld_dpatch, B, P, C, accumulate);
  return lc2is_check_launch();
}

extern "C" int lc2is_text_embed_fwd(const int64_t* ids, const float* tok, const float* pos, float* x, int ldx,
                                    int B, int L, int C, int vocab, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!ids || !tok || !pos || !x) return LC2IS_ERR_NULL;
  if (B <= 0 || L <= 0 || C <= 0 || C % 4 || vocab <= 0 || ldx < C || ldx % 4) return LC2IS_ERR_SHAPE;
  hipLaunchKernelGGL(text_embed_fwd_kernel, dim3(ew_grid((size_t)B * L * C / 4)), dim3(256), 0, stream, ids, tok,
                     pos, x, ldx, B * L, L, C, vocab);
  return lc2is_check_launch();
}

extern "C" int lc2is_text_embed_bwd(const int64_t* ids, const float* dx, int ldx, float* dtok, float* dpos,
                                    int B, int L, int C, int vocab, int accumulate, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!ids || !dx || !dtok || !dpos) return LC2IS_ERR_NULL;
  if (B <= 0 || L <= 0 || C <= 0 || vocab <= 0 || ldx < C) return LC2IS_ERR_SHAPE;
  hipLaunchKernelGGL(text_embed_dtok_kernel, dim3(B * L), dim3(256), 0, stream, ids, dx, ldx, dtok, B * L, C, vocab);
  int rc = lc2is_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(text_embed_bwd_kernel, dim3(ew_grid((size_t)L * C)), dim3(256), 0, stream, dx, ldx, dpos, B, L, C,
                     accumulate);
  return lc2is_check_launch();
}

extern "C" int lc2is_rows_copy_f32(const float* src, int S_src, int src_off, float* dst_f32, void* dst_bf16,
                                   int S_dst, int dst_off, int B, int n, int C, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!src || (!dst_f32 && !dst_bf16)) return LC2IS_ERR_NULL;
  if (B <= 0 || n <= 0 || C <= 0 || C % 4 || src_off < 0 || dst_off < 0 || src_off + n > S_src ||
      dst_off + n > S_dst)
    return LC2IS_ERR_SHAPE;
  hipLaunchKernelGGL(rows_copy_kernel, dim3(ew_grid((size_t)B * n * C / 4)), dim3(256), 0, stream, src, S_src,
                     src_off, dst_f32, (bf16_t*)dst_bf16, S_dst, dst_off, B, n, C);
  return lc2is_check_launch();
}

extern "C" int lc2is_rows_copy_bf16(const void* src, int S_src, int src_off, float* dst_f32, void* dst_bf16,
                                    int S_dst, int dst_off, int B, int n, int C, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!src || (!dst_f32 && !dst_bf16)) return LC2IS_ERR_NULL;
  if (B <= 0 || n <= 0 || C <= 0 || C % 4 || src_off < 0 || dst_off < 0 || src_off + n > S_src ||
      dst_off + n > S_dst)
    return LC2IS_ERR_SHAPE;
  hipLaunchKernelGGL(rows_copy_bf16_kernel, dim3(ew_grid((size_t)B * n * C / 4)), dim3(256), 0, stream, (const bf16_t*)src,
                     S_src, src_off, dst_f32, (bf16_t*)dst_bf16, S_dst, dst_off, B, n, C);
  return lc2is_check_launch();
}

extern "C" int lc2is_sgd_step(float* params, const float* grads, float* momentum_buf, size_t n, float lr,
                              float momentum, float weight_decay, float grad_scale, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!params || !grads) return LC2IS_ERR_NULL;
  if (n == 0 || n % 4) return LC2IS_ERR_SHAPE;
  hipLaunchKernelGGL(sgd_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, stream, params, grads, momentum_buf, n / 4,
                     lr, momentum, weight_decay, grad_scale);
  return lc2is_check_launch();
}

extern "C" int lc2is_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n,
                                float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                float grad_scale, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!params || !grads || !exp_avg || !exp_avg_sq) return LC2IS_ERR_NULL;
  if (n == 0 || n % 4 || step < 1) return LC2IS_ERR_SHAPE;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, stream, params, grads, exp_avg,
                     exp_avg_sq, n / 4, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale);
  return lc2is_check_launch();
}
