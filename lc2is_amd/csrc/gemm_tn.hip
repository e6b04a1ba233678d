// dW[N,K] (fp32) = dY[M,N]^T · X[M,K]  — the weight gradient of every nn.Linear on the hot path
// (autograd of the calls listed in gemm_nt.hip; reference engine.py:100 loss.backward()).
//
// The reduction runs over M (tokens), the slow dimension of BOTH operands, so neither tile can be
// read as a k-contiguous MFMA fragment.  gfx950 answers that with ds_read_b64_tr_b16: tiles are staged
// row-major [64 tokens][128 columns] (coalesced 16-byte global loads, range-checked so the token tail
// reads as zero) into LDS with the 256-byte-row XOR swizzle  chunk ^= ((row&3)<<2)|((row>>2)&3),
// and each 16x16x32 bf16 MFMA fragment is two transposed reads (4 token rows x 16 columns per 16-lane
// group), conflict-free under that swizzle.  MFMA A = X^T (rows = k), B = dY (cols = n), so a lane's 4
// accumulator registers are 4 consecutive k of one n: 16-byte fp32 stores into dW[n][k..k+3].
// Occupancy: the output is only (N/128)x(K/128) tiles, so M is cut into `splits` ranges; each writes an
// fp32 slab and a second launch sums the slabs in a fixed order (bitwise reproducible, no atomics).
#include "common.h"
#include "lc2is_hip.h"
#include <cstdlib>
#include <mutex>
#include <vector>
#include <type_traits>

namespace {

constexpr int TN_BN = 128, TN_BK = 128, TN_BM = 64;
constexpr int TN_TILE_BYTES = TN_BM * 256;  // one operand tile: 64 rows x 256 B
constexpr int TN_STAGE = 2 * TN_TILE_BYTES;

struct TnPlan {
  int ntn, ntk, splits, chunk;
  int big;  // 1: 256x256 LDS-DMA kernel, 0: 128x128 register-staged kernel
};

inline int tn_forced_cfg() {  // LC2IS_GEMM_TN_CFG=1|2 pins the kernel choice (tests / tools); default 0 = by shape
  static int cfg = -1;
  if (cfg < 0) {
    const char* e = getenv("LC2IS_GEMM_TN_CFG");
    cfg = e ? atoi(e) : 0;
  }
  return cfg;
}

// `ncu`: the CU budget, read ONCE per call by the entry point (a planner that re-read the process-wide atomic could see two values)
inline TnPlan tn_plan(int M, int N, int K, int ncu) {
  TnPlan p;
  p.big = 0;
  if (N % 256 == 0 && K % 256 == 0 && tn_forced_cfg() != 1) {
    const int ntn = N / 256, ntk = K / 256, tiles = ntn * ntk;
    int splits = (ncu + tiles / 2) / tiles;   // about one block per CU of the budget (common.h)
    const int max_splits = (M + 511) / 512;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (tiles * splits >= 128 || tn_forced_cfg() == 2) {
      int chunk = (M + splits - 1) / splits;
      chunk = (chunk + TN_BM - 1) / TN_BM * TN_BM;
      p.ntn = ntn; p.ntk = ntk; p.chunk = chunk;
      p.splits = (M + chunk - 1) / chunk;
      p.big = 1;
      return p;
    }
  }
  p.ntn = (N + TN_BN - 1) / TN_BN;
  p.ntk = (K + TN_BK - 1) / TN_BK;
  const int tiles = p.ntn * p.ntk;
  int splits = 1024 / tiles;
  // at least 8 steps of 64 rows per block; once the grid covers the chip twice over (>= 512 blocks) at least 32, so
  // that a few-tile gradient over hundreds of thousands of rows (Swin stage 1) is not all prologue and slab traffic
  int max_splits = (M + 511) / 512;
  const int long_splits = (M + 2047) / 2048;
  if ((long)tiles * long_splits >= 512) max_splits = long_splits;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int chunk = (M + splits - 1) / splits;
  chunk = (chunk + TN_BM - 1) / TN_BM * TN_BM;
  p.splits = (M + chunk - 1) / chunk;
  p.chunk = chunk;
  return p;
}

__device__ __forceinline__ int tn_swz(int row, int ch) {
  return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
}

__device__ __forceinline__ bf16x8_t tr_frag(const char* tile, int addr_lo, int addr_hi) {
  // two transposed 4x16 block reads -> the 8 reduction-dim elements of one 16x16x32 fragment
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4_t*)LDS_PTR(tile + addr_lo));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4_t*)LDS_PTR(tile + addr_hi));
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

__device__ __forceinline__ void tn_small_body(const bf16_t* __restrict__ dY, int ldy,
                                                       const bf16_t* __restrict__ X, int ldx, float* out,
                                                       int ldo, size_t split_stride, float* bias_out,
                                                       size_t bias_split_stride, int M, int N, int K,
                                                       int ntn, int ntk, int chunk, int accumulate, int blk_x, char* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wk = wid >> 1, wn = wid & 1;
  const int tiles = ntn * ntk;
  const int split = blk_x / tiles;
  const int tile = blk_x % tiles;
  const int n0 = (tile / ntk) * TN_BN, k0 = (tile % ntk) * TN_BK;
  const int m_begin = split * chunk;
  int m_end = m_begin + chunk;
  if (m_end > M) m_end = M;
  const int nsteps = (m_end - m_begin + TN_BM - 1) / TN_BM;

  // rows >= m_end must read as zero: the descriptor ends at row m_end (the next split owns the rest)
  const __amdgpu_buffer_rsrc_t rsY = make_rsrc(dY, (unsigned)m_end * (unsigned)ldy * 2u);
  const __amdgpu_buffer_rsrc_t rsX = make_rsrc(X, (unsigned)m_end * (unsigned)ldx * 2u);

  // staging: thread handles 4 chunks per operand: chunk c -> row c>>4, 16-byte column chunk c&15
  int y_goff[4], x_goff[4], st_lds[4];
  bool y_ok[4], x_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + i * 256, row = c >> 4, ch = c & 15;
    y_ok[i] = (n0 + ch * 8) < N;
    x_ok[i] = (k0 + ch * 8) < K;
    y_goff[i] = ((m_begin + row) * ldy + n0 + ch * 8) * 2;
    x_goff[i] = ((m_begin + row) * ldx + k0 + ch * 8) * 2;
    st_lds[i] = tn_swz(row, ch);
  }

  // transposed-read addresses (bytes inside a tile), per 32-row sub-step s and sub-tile t:
  // group g = lane>>4 owns reduction rows 8g..8g+7; lane 4q+p of the group addresses row 8g+q (+4),
  // columns 16*t' + 4p .. +3  ->  chunk = 2*t' + (p>>1), byte 8*(p&1) inside the chunk.
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  int xa_lo[2][4], xa_hi[2][4], ya_lo[2][4], ya_hi[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int r_lo = 32 * s + 8 * g + q, r_hi = r_lo + 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int chx = (wk * 64 + t * 16) / 8 + (pp >> 1);
      const int chy = (wn * 64 + t * 16) / 8 + (pp >> 1);
      xa_lo[s][t] = tn_swz(r_lo, chx) + 8 * (pp & 1);
      xa_hi[s][t] = tn_swz(r_hi, chx) + 8 * (pp & 1);
      ya_lo[s][t] = tn_swz(r_lo, chy) + 8 * (pp & 1);
      ya_hi[s][t] = tn_swz(r_hi, chy) + 8 * (pp & 1);
    }
  }

  f32x4_t acc[4][4];  // [k sub-tile][n sub-tile]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // fused bias gradient: db[n] = sum_m dY[m][n] = (ones^T · dY), one extra MFMA per n sub-tile with an all-ones
  // A fragment, on the dY fragments already in registers; only the k-tile-0 blocks' wk == 0 waves do it
  const bool do_cs = (bias_out != nullptr) && (tile % ntk == 0) && (wk == 0);
  f32x4_t cs[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) cs[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const s16x8_t ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s);

  i32x4_t ry[4], rx[4];
  auto gload = [&](int step) {
    const int yb = step * TN_BM * ldy * 2, xb = step * TN_BM * ldx * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ry[i] = __builtin_amdgcn_raw_buffer_load_b128(rsY, y_ok[i] ? y_goff[i] + yb : -1, 0, 0);
      rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, x_ok[i] ? x_goff[i] + xb : -1, 0, 0);
    }
  };
  auto lstore = [&](char* stage) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(i32x4_t*)(stage + st_lds[i]) = ry[i];
      *(i32x4_t*)(stage + TN_TILE_BYTES + st_lds[i]) = rx[i];
    }
  };

  if (nsteps > 0) {
    gload(0);
    lstore(smem);
  }
  __syncthreads();

  // the loop is instantiated twice (hand-unswitched on do_cs) so the common path carries no branch
  auto mainloop = [&](auto cs_tag) {
    constexpr bool CS = decltype(cs_tag)::value;
    for (int st = 0; st < nsteps; ++st) {
      const char* cur = smem + (st & 1) * TN_STAGE;
      char* nxt = smem + ((st + 1) & 1) * TN_STAGE;
      const bool more = (st + 1) < nsteps;
      if (more) gload(st + 1);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8_t xf[4], yf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          yf[t] = tr_frag(cur, ya_lo[s][t], ya_hi[s][t]);
          xf[t] = tr_frag(cur + TN_TILE_BYTES, xa_lo[s][t], xa_hi[s][t]);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[a], yf[b], acc[a][b], 0, 0, 0);
        if constexpr (CS) {
#pragma unroll
          for (int b = 0; b < 4; ++b) cs[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yf[b], cs[b], 0, 0, 0);
        }
      }
      if (more) lstore(nxt);
      __syncthreads();
    }
  };
  if (do_cs) mainloop(std::true_type{}); else mainloop(std::false_type{});

  float* o = out + (size_t)split * split_stride;
  if (do_cs && lane < 16) {  // every row of cs[b] holds the column sums; lanes 0..15, register 0 = row 0
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int n = n0 + wn * 64 + b * 16 + lane;
      if (n < N) {
        float* dst = bias_out + (size_t)split * bias_split_stride + n;
        *dst = (accumulate ? *dst : 0.f) + cs[b][0];
      }
    }
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int n = n0 + wn * 64 + b * 16 + (lane & 15);
    if (n >= N) continue;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int k = k0 + wk * 64 + a * 16 + g * 4;
      if (k >= K) continue;
      float* dst = o + (size_t)n * ldo + k;
      f32x4_t v = acc[a][b];
      if (accumulate) v += *(const f32x4_t*)dst;
      *(f32x4_t*)dst = v;
    }
  }
}

__global__ __launch_bounds__(256) void gemm_tn_kernel(const bf16_t* __restrict__ dY, int ldy,
                                                       const bf16_t* __restrict__ X, int ldx, float* out,
                                                       int ldo, size_t split_stride, float* bias_out,
                                                       size_t bias_split_stride, int M, int N, int K,
                                                       int ntn, int ntk, int chunk, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  tn_small_body(dY, ldy, X, ldx, out, ldo, split_stride, bias_out, bias_split_stride, M, N, K, ntn, ntk, chunk, accumulate,
                (int)blockIdx.x, smem);
}

// ---- 256x256 LDS-DMA variant for the large weight gradients --------------------------------------------------
// Same scheme on a 256(n) x 256(k) output tile with 8 waves (2 along k x 4 along n, 128x64 each: 24 transposed
// b64 reads per 32 MFMAs instead of 16 per 16), fed by LDS-DMA: a stage is 64 token rows x (512 B of dY | 512 B of
// X), a wave instruction moves two rows (1 KiB).  The row swizzle is applied on the GLOBAL side (lane l fetches
// chunk (l&31) ^ swz(row) of its row), so element (row, chunk) sits at row*512 + ((chunk ^ swz(row)) << 4): the
// low four chunk bits follow the 128x128 kernel's conflict-free rule, bit 4 is untouched.  Token rows past the
// split's end are out of the buffer descriptor's range and land in LDS as zeros.
constexpr int TD_BN = 256, TD_BK = 256;
constexpr int TD_TILE = TN_BM * 512;    // one operand tile: 64 rows x 512 B
constexpr int TD_STAGE = 2 * TD_TILE;   // 64 KiB

__device__ __forceinline__ int td_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ void td_dma(const bf16_t* dY, unsigned y_bytes, const bf16_t* X, unsigned x_bytes, char* stage,
                                       int wid, const int* y_goff, const int* x_goff, int ystep, int xstep) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the kernel stub; it has no LDS-DMA builtin
  const __amdgpu_buffer_rsrc_t rsY = make_rsrc(dY, y_bytes);
  const __amdgpu_buffer_rsrc_t rsX = make_rsrc(X, x_bytes);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, LDS_PTR(stage + (wid * 4 + j) * 1024), 16, y_goff[j], ystep, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, LDS_PTR(stage + TD_TILE + (wid * 4 + j) * 1024), 16, x_goff[j], xstep,
                                             0, 0);
#endif
}

__device__ __forceinline__ void tn_dma_body(const bf16_t* __restrict__ dY, int ldy, const bf16_t* __restrict__ X, int ldx,
                                            float* out, int ldo, size_t split_stride, float* bias_out,
                                            size_t bias_split_stride, int M, int N, int K, int ntn, int ntk, int chunk,
                                            int accumulate, int block, char* smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wid >> 2, wn = wid & 3;
  const int tiles = ntn * ntk;
  const int split = block / tiles;
  const int tile = block % tiles;
  const int n0 = (tile / ntk) * TD_BN, k0 = (tile % ntk) * TD_BK;
  const int m_begin = split * chunk;
  int m_end = m_begin + chunk;
  if (m_end > M) m_end = M;
  const int nsteps = (m_end - m_begin + TN_BM - 1) / TN_BM;
  const unsigned y_bytes = (unsigned)m_end * (unsigned)ldy * 2u, x_bytes = (unsigned)m_end * (unsigned)ldx * 2u;

  // DMA: piece 4*wid + j = token rows 2*(4*wid+j), +1; lane -> (row, 16-byte position), source chunk swizzled
  int y_goff[4], x_goff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 2 * (wid * 4 + j) + (lane >> 5);
    const int ch = (lane & 31) ^ td_swz(row);
    y_goff[j] = ((m_begin + row) * ldy + n0 + ch * 8) * 2;
    x_goff[j] = ((m_begin + row) * ldx + k0 + ch * 8) * 2;
  }

  // transposed-read bases: group g owns reduction rows 8g..8g+7 of a 32-row sub-step, lane 4q+p of the group
  // addresses row 8g+q (lo) / 8g+q+4 (hi), 16-byte chunk 2*t' + (p>>1), byte 8*(p&1).  t' only moves chunk
  // bits 1..3 = address bits 5..7, so the address of sub-tile t is  base ^ (t << 5).
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int r_lo = 8 * g + q, r_hi = r_lo + 4;
  const int x_lo = TD_TILE + r_lo * 512 + (((wk * 16 + (pp >> 1)) ^ td_swz(r_lo)) << 4) + 8 * (pp & 1);
  const int x_hi = TD_TILE + r_hi * 512 + (((wk * 16 + (pp >> 1)) ^ td_swz(r_hi)) << 4) + 8 * (pp & 1);
  const int y_lo = r_lo * 512 + (((wn * 8 + (pp >> 1)) ^ td_swz(r_lo)) << 4) + 8 * (pp & 1);
  const int y_hi = r_hi * 512 + (((wn * 8 + (pp >> 1)) ^ td_swz(r_hi)) << 4) + 8 * (pp & 1);

  f32x4_t acc[8][4];  // [k sub-tile][n sub-tile]
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const bool do_cs = (bias_out != nullptr) && (tile % ntk == 0) && (wk == 0);
  f32x4_t cs[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) cs[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const s16x8_t ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s);

  if (nsteps > 0) td_dma(dY, y_bytes, X, x_bytes, smem, wid, y_goff, x_goff, 0, 0);
  __syncthreads();  // vmcnt(0) + barrier: stage 0 has landed

  auto mainloop = [&](auto cs_tag) {
    constexpr bool CS = decltype(cs_tag)::value;
    for (int st = 0; st < nsteps; ++st) {
      const char* cur = smem + (st & 1) * TD_STAGE;
      if (st + 1 < nsteps)
        td_dma(dY, y_bytes, X, x_bytes, smem + ((st + 1) & 1) * TD_STAGE, wid, y_goff, x_goff,
               (st + 1) * TN_BM * ldy * 2, (st + 1) * TN_BM * ldx * 2);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8_t xf[8], yf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) yf[t] = tr_frag(cur + s * 16384, y_lo ^ (t << 5), y_hi ^ (t << 5));
#pragma unroll
        for (int t = 0; t < 8; ++t) xf[t] = tr_frag(cur + s * 16384, x_lo ^ (t << 5), x_hi ^ (t << 5));
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int a = 0; a < 8; ++a)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[a], yf[b], acc[a][b], 0, 0, 0);
        if constexpr (CS) {
#pragma unroll
          for (int b = 0; b < 4; ++b) cs[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yf[b], cs[b], 0, 0, 0);
        }
      }
      __syncthreads();  // reads of `cur` done; the next stage's DMA retired (vmcnt(0)) and published
    }
  };
  if (do_cs) mainloop(std::true_type{}); else mainloop(std::false_type{});

  float* o = out + (size_t)split * split_stride;
  if (do_cs && lane < 16) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int n = n0 + wn * 64 + b * 16 + lane;
      float* dst = bias_out + (size_t)split * bias_split_stride + n;
      *dst = (accumulate ? *dst : 0.f) + cs[b][0];
    }
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int n = n0 + wn * 64 + b * 16 + (lane & 15);
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const int k = k0 + wk * 128 + a * 16 + g * 4;
      float* dst = o + (size_t)n * ldo + k;
      f32x4_t v = acc[a][b];
      if (accumulate) v += *(const f32x4_t*)dst;
      *(f32x4_t*)dst = v;
    }
  }
}

__global__ __launch_bounds__(512, 2) void gemm_tn_dma_kernel(const bf16_t* __restrict__ dY, int ldy,
                                                              const bf16_t* __restrict__ X, int ldx, float* out, int ldo,
                                                              size_t split_stride, float* bias_out,
                                                              size_t bias_split_stride, int M, int N, int K, int ntn,
                                                              int ntk, int chunk, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  tn_dma_body(dY, ldy, X, ldx, out, ldo, split_stride, bias_out, bias_split_stride, M, N, K, ntn, ntk, chunk, accumulate,
              blockIdx.x, smem);
}

// ---- grouped launch: the weight gradients of one transformer layer in ONE grid ---------------------------------------
// A ViT layer has six weight gradients over the same tokens (q, k, v, out-proj: 9 tiles each; fc1, fc2: 36 each).  Launched
// one by one each fills the chip only by cutting M into 7-28 slabs, so every block is short (19-74 steps) against a fixed
// prologue + 256-KiB epilogue, and each launch drags its own ordered-reduce launch.  Grouped, the 108 tiles need only a
// 2-way M split to give every CU a long block: fewer, longer blocks, a quarter of the slab traffic, one reduce launch.
constexpr int TG_MAX = 16;
struct TnGroupProb {
  const bf16_t* dY; const bf16_t* X; float* out; float* bias_out; float* dW; float* db;
  size_t split_stride, bias_split_stride;
  int ldy, ldx, ldo, ldw, M, N, K, ntn, ntk, chunk, splits, accumulate;
  int blk_end;      // exclusive end of this problem's block range in the main grid
  int red_end;      // ... and in the reduce grid
  int red_blocks;   // slab blocks of the reduce grid (the rest of the problem's range sums the bias partials)
};
struct TnGroup {
  int n, remap;
  TnGroupProb p[TG_MAX];
};

// More than TG_MAX problems (the weight gradients of a whole tower in one grid): the descriptors live in device memory
// (front of the workspace, uploaded per call); the block-range ends sit in their own dense arrays so that the search
// touches a few cache lines.
constexpr int TG_TBL_MAX = LC2IS_TN_GROUP_MAX;
struct TnGroupTbl {
  int n, pad_[3];
  int blk_end[TG_TBL_MAX];
  int red_end[TG_TBL_MAX];
  TnGroupProb p[TG_TBL_MAX];
};
constexpr size_t TG_TBL_BYTES = (sizeof(TnGroupTbl) + 255) / 256 * 256;

// XCD-aware block order (t->pad_[0] / g.remap): blocks b, b+8, ... share an XCD and its L2, but neighbouring LOGICAL blocks are the
// tiles of one problem, which share operand panels (a 256-column panel of dY serves K/256 tiles, one of X serves N/256): with the
// plain order every tile streamed both of its panels from the Infinity Cache / HBM by itself — rocprofv3 FETCH_SIZE 38 GB per
// whole-tower launch, 7.3 TB/s over its 5.2 ms.
__global__ __launch_bounds__(512, 2) void gemm_tn_grouped_tbl_kernel(const TnGroupTbl* __restrict__ t) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int i = 0, begin = 0;
  const int n = t->n;
  const int bid = t->pad_[0] ? xcd_round_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
#pragma unroll 1
  while (i < n - 1 && bid >= t->blk_end[i]) { begin = t->blk_end[i]; ++i; }
  const TnGroupProb* q = &t->p[i];
  tn_dma_body(q->dY, q->ldy, q->X, q->ldx, q->out, q->ldo, q->split_stride, q->bias_out, q->bias_split_stride, q->M, q->N,
              q->K, q->ntn, q->ntk, q->chunk, q->splits == 1 ? q->accumulate : 0, bid - begin, smem);
}

// groups whose N / K are not all multiples of 256 (the Swin blocks: 96 .. 768 channels and their 4x MLPs) run the 128x128
// register-staged body of gemm_tn_kernel per block instead: one grid + one ordered reduce per layer, not two launches per weight
__global__ __launch_bounds__(256) void gemm_tn_grouped_small_kernel(TnGroup g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int i = 0, begin = 0;
#pragma unroll 1
  while (i < g.n - 1 && (int)blockIdx.x >= g.p[i].blk_end) { begin = g.p[i].blk_end; ++i; }
  const TnGroupProb& q = g.p[i];
  tn_small_body(q.dY, q.ldy, q.X, q.ldx, q.out, q.ldo, q.split_stride, q.bias_out, q.bias_split_stride, q.M, q.N, q.K, q.ntn,
                q.ntk, q.chunk, q.splits == 1 ? q.accumulate : 0, (int)blockIdx.x - begin, smem);
}

__global__ __launch_bounds__(256) void gemm_tn_grouped_small_tbl_kernel(const TnGroupTbl* __restrict__ t) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int i = 0, begin = 0;
  const int n = t->n;
#pragma unroll 1
  while (i < n - 1 && (int)blockIdx.x >= t->blk_end[i]) { begin = t->blk_end[i]; ++i; }
  const TnGroupProb* q = &t->p[i];
  tn_small_body(q->dY, q->ldy, q->X, q->ldx, q->out, q->ldo, q->split_stride, q->bias_out, q->bias_split_stride, q->M, q->N,
                q->K, q->ntn, q->ntk, q->chunk, q->splits == 1 ? q->accumulate : 0, (int)blockIdx.x - begin, smem);
}

__global__ __launch_bounds__(512, 2) void gemm_tn_grouped_kernel(TnGroup g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int i = 0, begin = 0;
  const int bid = g.remap ? xcd_round_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
#pragma unroll 1
  while (i < g.n - 1 && bid >= g.p[i].blk_end) { begin = g.p[i].blk_end; ++i; }
  const TnGroupProb& q = g.p[i];
  tn_dma_body(q.dY, q.ldy, q.X, q.ldx, q.out, q.ldo, q.split_stride, q.bias_out, q.bias_split_stride, q.M, q.N, q.K, q.ntn,
              q.ntk, q.chunk, q.splits == 1 ? q.accumulate : 0, bid - begin, smem);
}

__device__ __forceinline__ void slab_reduce_grouped_body(const TnGroupProb& q, int b);

__global__ __launch_bounds__(256) void slab_reduce_grouped_kernel(TnGroup g) {
  int i = 0, begin = 0;
#pragma unroll 1
  while (i < g.n - 1 && (int)blockIdx.x >= g.p[i].red_end) { begin = g.p[i].red_end; ++i; }
  slab_reduce_grouped_body(g.p[i], (int)blockIdx.x - begin);
}

__global__ __launch_bounds__(256) void slab_reduce_grouped_tbl_kernel(const TnGroupTbl* __restrict__ t) {
  int i = 0, begin = 0;
  const int n = t->n;
#pragma unroll 1
  while (i < n - 1 && (int)blockIdx.x >= t->red_end[i]) { begin = t->red_end[i]; ++i; }
  slab_reduce_grouped_body(t->p[i], (int)blockIdx.x - begin);
}

__device__ __forceinline__ void slab_reduce_grouped_body(const TnGroupProb& q, int b) {
  if (q.splits <= 1) return;
  if (b >= q.red_blocks) {                     // fused bias-gradient partials, one column per thread, fixed order
    const int n = (b - q.red_blocks) * 256 + threadIdx.x;
    if (q.db && n < q.N) {
      float t = 0.f;
      for (int sp = 0; sp < q.splits; ++sp) t += q.bias_out[(size_t)sp * q.bias_split_stride + n];
      q.db[n] = q.accumulate ? q.db[n] + t : t;
    }
    return;
  }
  const int K4 = q.K >> 2;
  const size_t total = (size_t)q.N * K4;
  for (size_t idx = (size_t)b * 256 + threadIdx.x; idx < total; idx += (size_t)q.red_blocks * 256) {
    const int n = (int)(idx / K4), k4 = (int)(idx % K4);
    float4 s4 = *(reinterpret_cast<const float4*>(q.out + (size_t)n * q.K) + k4);
    for (int sp = 1; sp < q.splits; ++sp) {
      const float4 v = *reinterpret_cast<const float4*>(q.out + sp * q.split_stride + (size_t)n * q.K + 4 * k4);
      s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
    }
    float4* dst = reinterpret_cast<float4*>(q.dW + (size_t)n * q.ldw) + k4;
    if (q.accumulate) {
      const float4 d = *dst;
      s4.x += d.x; s4.y += d.y; s4.z += d.z; s4.w += d.w;
    }
    *dst = s4;
  }
}

// blocks [0, gridDim.x - bias_blocks) sum the dW slabs; the last bias_blocks blocks sum the fused bias-gradient
// partials (one column per thread, fixed order) so the weight gradient needs two launches, not three
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ ws, int splits,
                                                           size_t split_stride, float* out, int ldo, int N,
                                                           int K, int accumulate, const float* __restrict__ bias_ws,
                                                           float* db, int bias_blocks) {
  const int slab_blocks = gridDim.x - bias_blocks;
  if ((int)blockIdx.x >= slab_blocks) {
    const int n = (blockIdx.x - slab_blocks) * 256 + threadIdx.x;
    if (n < N) {
      float t = 0.f;
      for (int sp = 0; sp < splits; ++sp) t += bias_ws[(size_t)sp * N + n];
      db[n] = accumulate ? db[n] + t : t;
    }
    return;
  }
  const int K4 = K >> 2;
  const size_t total = (size_t)N * K4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)slab_blocks * 256) {
    const int n = (int)(i / K4), k4 = (int)(i % K4);
    const float4* src = reinterpret_cast<const float4*>(ws + (size_t)n * K) + k4;
    float4 s = *src;
    for (int sp = 1; sp < splits; ++sp) {
      const float4 v = *reinterpret_cast<const float4*>(ws + sp * split_stride + (size_t)n * K + 4 * k4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float4* dst = reinterpret_cast<float4*>(out + (size_t)n * ldo) + k4;
    if (accumulate) {
      const float4 d = *dst;
      s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
    }
    *dst = s;
  }
}

// ---- bias gradient: column sums of a bf16 [M,N] matrix -------------------------------------------------
constexpr int CS_ROWBLOCKS = 128;

__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ dY, int ldy, float* ws, int M,
                                                      int N) {
  // block (bx, by): 64 column groups of 8 (lane) x 4 row lanes (wave); rows by*4+wave, step 4*gridDim.y
  __shared__ float red[4][64][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cg = blockIdx.x * 64 + lane;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (cg * 8 < N) {
    for (int row = blockIdx.y * 4 + wave; row < M; row += gridDim.y * 4) {
      const uint4 v = *reinterpret_cast<const uint4*>(dY + (size_t)row * ldy + cg * 8);
      s[0] += bf16_to_f32((bf16_t)(v.x & 0xffff)); s[1] += bf16_to_f32((bf16_t)(v.x >> 16));
      s[2] += bf16_to_f32((bf16_t)(v.y & 0xffff)); s[3] += bf16_to_f32((bf16_t)(v.y >> 16));
      s[4] += bf16_to_f32((bf16_t)(v.z & 0xffff)); s[5] += bf16_to_f32((bf16_t)(v.z >> 16));
      s[6] += bf16_to_f32((bf16_t)(v.w & 0xffff)); s[7] += bf16_to_f32((bf16_t)(v.w >> 16));
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[wave][lane][j] = s[j];
  __syncthreads();
  if (wave == 0 && cg * 8 < N) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      ws[(size_t)blockIdx.y * N + cg * 8 + j] =
          (red[0][lane][j] + red[1][lane][j]) + (red[2][lane][j] + red[3][lane][j]);
  }
}

__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ ws, int nparts, int N,
                                                             float* db, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  float s = 0.f;
  for (int p = 0; p < nparts; ++p) s += ws[(size_t)p * N + c];
  db[c] = accumulate ? db[c] + s : s;
}

inline int colsum_parts(int M) {
  int p = (M + 3) / 4;
  return p > CS_ROWBLOCKS ? CS_ROWBLOCKS : p;
}

}  // namespace

static size_t tn_plan_bytes(const TnPlan& p, int N, int K) {
  return p.splits > 1 ? (size_t)p.splits * N * ((size_t)K + 1) * sizeof(float) : 0;
}

extern "C" size_t lc2is_gemm_tn_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const TnPlan p = tn_plan(M, N, K, lc2is_ncu());
  return tn_plan_bytes(p, N, K);
}

extern "C" int lc2is_gemm_tn_bf16(const void* dY, int ldy, const void* X, int ldx, float* dW, int ldw, float* db,
                                  int M, int N, int K, int accumulate, void* workspace, size_t workspace_bytes,
                                  lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!dY || !X || !dW) return LC2IS_ERR_NULL;
  if (M <= 0 || N <= 0 || K <= 0 || N % 8 || K % 8) return LC2IS_ERR_SHAPE;
  if (ldy < N || ldx < K || ldw < K || ldy % 8 || ldx % 8 || ldw % 4) return LC2IS_ERR_SHAPE;
  if ((double)(M + 64) * ldy * 2.0 >= 2147483648.0 || (double)(M + 64) * ldx * 2.0 >= 2147483648.0)
    return LC2IS_ERR_UNSUPPORTED;
  const TnPlan p = tn_plan(M, N, K, lc2is_ncu());   // ONE read of the CU budget: the launch and its workspace need come from this plan
  const size_t need = tn_plan_bytes(p, N, K);
  if (need && (!workspace || workspace_bytes < need)) return LC2IS_ERR_WORKSPACE;
  static DevOnce attr_set;
  if (attr_set.need()) {
    if (hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * TN_STAGE) != hipSuccess)
      return LC2IS_ERR_LAUNCH;
    attr_set.done();
  }
  static DevOnce attr_big_set;
  if (p.big && attr_big_set.need()) {
    if (hipFuncSetAttribute((const void*)gemm_tn_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * TD_STAGE) != hipSuccess)
      return LC2IS_ERR_LAUNCH;
    attr_big_set.done();
  }
  const int grid = p.ntn * p.ntk * p.splits;
  auto launch = [&](float* o, int ldo, size_t o_stride, float* b, size_t b_stride, int acc_flag) {
    if (p.big)
      hipLaunchKernelGGL(gemm_tn_dma_kernel, dim3(grid), dim3(512), 2 * TD_STAGE, stream, (const bf16_t*)dY, ldy,
                         (const bf16_t*)X, ldx, o, ldo, o_stride, b, b_stride, M, N, K, p.ntn, p.ntk, p.chunk, acc_flag);
    else
      hipLaunchKernelGGL(gemm_tn_kernel, dim3(grid), dim3(256), 2 * TN_STAGE, stream, (const bf16_t*)dY, ldy,
                         (const bf16_t*)X, ldx, o, ldo, o_stride, b, b_stride, M, N, K, p.ntn, p.ntk, p.chunk, acc_flag);
  };
  if (p.splits == 1) {
    launch(dW, ldw, (size_t)0, db, (size_t)0, accumulate);
    return lc2is_check_launch();
  }
  float* bias_ws = db ? (float*)workspace + (size_t)p.splits * N * K : nullptr;
  launch((float*)workspace, K, (size_t)N * K, bias_ws, (size_t)N, 0);
  int rc = lc2is_check_launch();
  if (rc) return rc;
  const size_t total4 = (size_t)N * K / 4;
  int rgrid = (int)((total4 + 255) / 256);
  if (rgrid > 2048) rgrid = 2048;
  const int bias_blocks = db ? (N + 255) / 256 : 0;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(rgrid + bias_blocks), dim3(256), 0, stream, (const float*)workspace,
                     p.splits, (size_t)N * K, dW, ldw, N, K, accumulate, (const float*)bias_ws, db, bias_blocks);
  return lc2is_check_launch();
}

// ---- grouped weight gradients -------------------------------------------------------------------------------------
namespace {
struct TgPlan {
  int splits[TG_TBL_MAX];   // M-split count per problem
  int order[TG_TBL_MAX];    // problem indices in grid order (split problems last: their short blocks fill the last round)
  size_t ws_floats;         // slab floats (after the descriptor table when n > TG_MAX)
  bool small;               // some N or K is not a multiple of 256: 128x128 tiles, register-staged body
};

inline int tg_valid(const lc2is_tn_problem* pr, int n) {
  if (!pr || n <= 0 || n > TG_TBL_MAX) return LC2IS_ERR_UNSUPPORTED;
  for (int i = 0; i < n; ++i) {
    const lc2is_tn_problem& q = pr[i];
    if (!q.dY || !q.X || !q.dW) return LC2IS_ERR_NULL;
    if (q.M <= 0 || q.N <= 0 || q.K <= 0 || q.N % 8 || q.K % 8) return LC2IS_ERR_UNSUPPORTED;
    if (q.ldy < q.N || q.ldx < q.K || q.ldw < q.K || q.ldy % 8 || q.ldx % 8 || q.ldw % 4) return LC2IS_ERR_SHAPE;
    if ((double)(q.M + 64) * q.ldy * 2.0 >= 2147483648.0 || (double)(q.M + 64) * q.ldx * 2.0 >= 2147483648.0)
      return LC2IS_ERR_UNSUPPORTED;
  }
  return LC2IS_OK;
}

// Cost in units of one 64-row step of a 256x256 tile (~1.65 us); 13 steps ~ prologue + 256-KiB epilogue of a block; one
// block per CU, so time = rounds x block length.  Two candidate plans:
//  (uniform) one M-split count for the whole group: rounds x (longest block + 13) + slab traffic — the right shape for a
//            single layer (108 tiles: 2 splits, 216 blocks in one round);
//  (tail)    every tile a full-length block except a few small problems that are split so finely that their blocks fill
//            the last, partly empty round (a whole tower, 1296 tiles: 1278 full blocks in 5 rounds + 18 tiles x 14 splits)
//            — no slabs for the bulk and ~97 % of the CUs busy instead of 84 %.
inline void tg_plan(const lc2is_tn_problem* pr, int n, TgPlan& pl, const int ncu) {
  pl.small = false;
  for (int i = 0; i < n; ++i)
    if (pr[i].N % 256 || pr[i].K % 256) pl.small = true;
  if (pl.small) {
    // 128x128 tiles, two 256-thread blocks per CU (64 KB of LDS each): aim at ~1024 blocks for the group, at least 8 steps
    // of 64 rows per block (at least 32 once the grid is that full anyway: Swin stage 1 has few tiles over 260k rows)
    long tiles = 0;
    for (int i = 0; i < n; ++i) tiles += (long)((pr[i].N + TN_BN - 1) / TN_BN) * ((pr[i].K + TN_BK - 1) / TN_BK);
    pl.ws_floats = 0;
    for (int i = 0; i < n; ++i) {
      int sp = (int)((1024 + tiles / 2) / tiles);
      int max_sp = (pr[i].M + 511) / 512;
      const int long_sp = (pr[i].M + 2047) / 2048;
      if (tiles * long_sp >= 512) max_sp = long_sp;
      if (sp > max_sp) sp = max_sp;
      if (sp < 1) sp = 1;
      pl.splits[i] = sp;
      pl.order[i] = i;
      if (sp > 1) pl.ws_floats += (size_t)sp * pr[i].N * ((size_t)pr[i].K + 1);
    }
    return;
  }
  long tiles = 0;
  double wbytes = 0;
  int max_steps = 0, max_splits = 1 << 30;
  int ptiles[TG_TBL_MAX];
  for (int i = 0; i < n; ++i) {
    ptiles[i] = (pr[i].N / 256) * (pr[i].K / 256);
    tiles += ptiles[i];
    wbytes += 4.0 * pr[i].N * pr[i].K;
    const int steps = (pr[i].M + TN_BM - 1) / TN_BM;
    if (steps > max_steps) max_steps = steps;
    const int ms = (pr[i].M + 511) / 512;
    if (ms < max_splits) max_splits = ms;
  }
  if (max_splits > 32) max_splits = 32;
  if (max_splits < 1) max_splits = 1;
  const double slab_unit = 2.0 / 4.0e12 / 1.65e-6;   // steps per byte of slab written and read back
  int best = 1;
  double best_t = 1e300;
  for (int sp = 1; sp <= max_splits; ++sp) {
    const long blocks = tiles * sp;
    const double rounds = (double)((blocks + ncu - 1) / ncu);
    const double t = rounds * ((max_steps + sp - 1) / sp + 13.0) + (sp > 1 ? sp * wbytes * slab_unit : 0.0);
    if (t < best_t) { best_t = t; best = sp; }
  }
  for (int i = 0; i < n; ++i) { pl.splits[i] = best; pl.order[i] = i; }
  // (ncu: 256, or the budget set while another queue's kernels hold CUs: common.h)
  const int rem = (int)(tiles % ncu);
  if (n > 1 && tiles > ncu && rem != 0 && max_splits >= 2) {
    bool in_tail[TG_TBL_MAX] = {};
    int ts = 0;
    double tail_bytes = 0;
    while (ts < rem) {   // smallest problems first (later index on ties)
      int pick = -1;
      for (int i = 0; i < n; ++i)
        if (!in_tail[i] && (pick < 0 || ptiles[i] <= ptiles[pick])) pick = i;
      if (pick < 0) break;
      in_tail[pick] = true;
      ts += ptiles[pick];
      tail_bytes += 4.0 * pr[pick].N * pr[pick].K;
    }
    int sp = ts > 0 ? ncu / ts : 0;
    if (sp > max_splits) sp = max_splits;
    if (ts >= rem && sp >= 2) {
      const double rounds = (double)((tiles - ts + ncu - 1) / ncu);
      const double t = rounds * (max_steps + 13.0) + ((max_steps + sp - 1) / sp + 13.0) + sp * tail_bytes * slab_unit;
      if (t < best_t) {
        int k = 0;
        for (int i = 0; i < n; ++i) if (!in_tail[i]) { pl.order[k++] = i; pl.splits[i] = 1; }
        for (int i = 0; i < n; ++i) if (in_tail[i]) { pl.order[k++] = i; pl.splits[i] = sp; }
      }
    }
  }
  pl.ws_floats = 0;
  for (int i = 0; i < n; ++i)
    if (pl.splits[i] > 1) pl.ws_floats += (size_t)pl.splits[i] * pr[i].N * ((size_t)pr[i].K + 1);
}
}  // namespace

static std::mutex g_captured_mu;
static std::vector<void*> g_captured_tables;   // pinned descriptor-table images of captured grouped launches

extern "C" int lc2is_release_captured_tables(void) {
  std::lock_guard<std::mutex> lock(g_captured_mu);
  int n = 0;
  for (void*& p : g_captured_tables)   // the table never shrinks: marks held by other captured steps stay valid indices (their slots
    if (p) { (void)hipHostFree(p); p = nullptr; ++n; }   // are simply empty afterwards; a later capture appends, it never reuses a slot)
  return n;
}

// Per-graph ownership: the images registered between two marks belong to the graph captured in between.
extern "C" int lc2is_captured_tables_mark(void) {
  std::lock_guard<std::mutex> lock(g_captured_mu);
  return (int)g_captured_tables.size();
}

extern "C" int lc2is_release_captured_tables_range(int first, int last) {
  std::lock_guard<std::mutex> lock(g_captured_mu);
  if (first < 0) first = 0;
  if (last > (int)g_captured_tables.size()) last = (int)g_captured_tables.size();
  int n = 0;
  for (int i = first; i < last; ++i)
    if (g_captured_tables[i]) { (void)hipHostFree(g_captured_tables[i]); g_captured_tables[i] = nullptr; ++n; }   // (slots keep their index)
  return n;
}

extern "C" size_t lc2is_gemm_tn_grouped_workspace_bytes(const lc2is_tn_problem* problems, int n) {
  if (tg_valid(problems, n) != LC2IS_OK) return 0;
  TgPlan pl;
  tg_plan(problems, n, pl, lc2is_ncu());
  return pl.ws_floats * sizeof(float) + (n > TG_MAX ? TG_TBL_BYTES : 0);
}

extern "C" int lc2is_gemm_tn_grouped(const lc2is_tn_problem* problems, int n, void* workspace, size_t workspace_bytes,
                                     lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = tg_valid(problems, n);
  if (rc) return rc;
  TgPlan pl;
  tg_plan(problems, n, pl, lc2is_ncu());
  const bool tbl = n > TG_MAX;
  const size_t need = pl.ws_floats * sizeof(float) + (tbl ? TG_TBL_BYTES : 0);
  if (need && (!workspace || workspace_bytes < need)) return LC2IS_ERR_WORKSPACE;
  // Host image of the table: a ring of PINNED slots, each guarded by an event recorded behind its upload, so a slot is
  // never rewritten while an asynchronous H2D copy may still be reading it (whatever the runtime does with pageable memory).
  // Under stream capture the upload becomes a memcpy node that re-reads its host source at EVERY replay, so a captured
  // call gets a pinned image of its own that is never reused; it lives as long as its graph (lc2is_release_captured_tables).
  struct TblSlot { TnGroupTbl* host; hipEvent_t done; bool in_flight; };
  static thread_local TblSlot ring[4] = {};
  static thread_local unsigned ring_pos = 0;
  static thread_local TnGroupTbl small_tbl;     // <= TG_MAX problems travel as kernel arguments: plain host memory
  TblSlot* slot = nullptr;
  TnGroupTbl* captured_tbl = nullptr;
  if (tbl) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) return LC2IS_ERR_LAUNCH;
    if (cap != hipStreamCaptureStatusNone) {
      // (an allocation is an "unsafe" call under the default global capture mode: relax this thread's mode around it)
      hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
      if (hipThreadExchangeStreamCaptureMode(&mode) != hipSuccess) return LC2IS_ERR_LAUNCH;
      const hipError_t e = hipHostMalloc((void**)&captured_tbl, sizeof(TnGroupTbl), hipHostMallocDefault);
      if (hipThreadExchangeStreamCaptureMode(&mode) != hipSuccess || e != hipSuccess) return LC2IS_ERR_LAUNCH;
      {   // owned by the graph being captured: remembered so that lc2is_release_captured_tables() can free it with the graph
        std::lock_guard<std::mutex> lock(g_captured_mu);
        g_captured_tables.push_back(captured_tbl);
      }
    } else {
      slot = &ring[ring_pos++ & 3];
      if (!slot->host) {
        if (hipHostMalloc((void**)&slot->host, sizeof(TnGroupTbl), hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&slot->done, hipEventDisableTiming) != hipSuccess)
          return LC2IS_ERR_LAUNCH;
      }
      if (slot->in_flight && hipEventSynchronize(slot->done) != hipSuccess) return LC2IS_ERR_LAUNCH;
      slot->in_flight = false;
    }
  }
  TnGroupTbl& t = captured_tbl ? *captured_tbl : (tbl ? *slot->host : small_tbl);
  static const int xcd_order = !(getenv("LC2IS_TN_XCD") && atoi(getenv("LC2IS_TN_XCD")) == 0);
  t.n = n;
  t.pad_[0] = xcd_order;
  float* ws = (float*)((char*)workspace + (tbl ? TG_TBL_BYTES : 0));
  int blk = 0, red = 0;
  for (int k = 0; k < n; ++k) {
    const int i = pl.order[k];
    const lc2is_tn_problem& q = problems[i];
    TnGroupProb& d = t.p[k];
    d.dY = (const bf16_t*)q.dY; d.X = (const bf16_t*)q.X; d.dW = q.dW; d.db = q.db;
    d.ldy = q.ldy; d.ldx = q.ldx; d.ldw = q.ldw; d.M = q.M; d.N = q.N; d.K = q.K; d.accumulate = q.accumulate;
    d.ntn = pl.small ? (q.N + TN_BN - 1) / TN_BN : q.N / 256;
    d.ntk = pl.small ? (q.K + TN_BK - 1) / TN_BK : q.K / 256;
    int chunk = (q.M + pl.splits[i] - 1) / pl.splits[i];
    chunk = (chunk + TN_BM - 1) / TN_BM * TN_BM;
    d.chunk = chunk;
    d.splits = (q.M + chunk - 1) / chunk;
    if (pl.splits[i] > 1) {   // (slab space is reserved for the planned count even if the rounded chunk needs fewer)
      float* base = ws;
      ws += (size_t)pl.splits[i] * q.N * ((size_t)q.K + 1);
      if (d.splits > 1) {
        d.out = base; d.ldo = q.K; d.split_stride = (size_t)q.N * q.K;
        d.bias_out = q.db ? base + (size_t)d.splits * q.N * q.K : nullptr; d.bias_split_stride = (size_t)q.N;
      }
    }
    if (d.splits <= 1) {
      d.out = q.dW; d.ldo = q.ldw; d.split_stride = 0; d.bias_out = q.db; d.bias_split_stride = 0;
    }
    blk += d.ntn * d.ntk * d.splits;
    d.blk_end = blk;
    const size_t total4 = (size_t)q.N * q.K / 4;
    int rb = (int)((total4 + 255) / 256);
    if (rb > 1024) rb = 1024;
    d.red_blocks = d.splits > 1 ? rb : 0;
    red += d.splits > 1 ? rb + (q.db ? (q.N + 255) / 256 : 0) : 0;
    d.red_end = red;
    t.blk_end[k] = blk;
    t.red_end[k] = red;
  }
  static DevOnce attr_set;
  if (attr_set.need()) {
    if (hipFuncSetAttribute((const void*)gemm_tn_grouped_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * TD_STAGE) != hipSuccess ||
        hipFuncSetAttribute((const void*)gemm_tn_grouped_tbl_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * TD_STAGE) != hipSuccess ||
        hipFuncSetAttribute((const void*)gemm_tn_grouped_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * TN_STAGE) != hipSuccess ||
        hipFuncSetAttribute((const void*)gemm_tn_grouped_small_tbl_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * TN_STAGE) != hipSuccess)
      return LC2IS_ERR_LAUNCH;
    attr_set.done();
  }
  if (tbl) {
    if (hipMemcpyAsync(workspace, &t, sizeof(TnGroupTbl), hipMemcpyHostToDevice, stream) != hipSuccess) return LC2IS_ERR_LAUNCH;
    if (slot) {
      if (hipEventRecord(slot->done, stream) != hipSuccess) return LC2IS_ERR_LAUNCH;
      slot->in_flight = true;
    }
    const TnGroupTbl* dt = (const TnGroupTbl*)workspace;
    if (pl.small)
      hipLaunchKernelGGL(gemm_tn_grouped_small_tbl_kernel, dim3(blk), dim3(256), 2 * TN_STAGE, stream, dt);
    else
      hipLaunchKernelGGL(gemm_tn_grouped_tbl_kernel, dim3(blk), dim3(512), 2 * TD_STAGE, stream, dt);
    rc = lc2is_check_launch();
    if (rc || red == 0) return rc;
    hipLaunchKernelGGL(slab_reduce_grouped_tbl_kernel, dim3(red), dim3(256), 0, stream, dt);
    return lc2is_check_launch();
  }
  TnGroup g{};
  g.n = n;
  g.remap = xcd_order;
  for (int k = 0; k < n; ++k) g.p[k] = t.p[k];
  if (pl.small)
    hipLaunchKernelGGL(gemm_tn_grouped_small_kernel, dim3(blk), dim3(256), 2 * TN_STAGE, stream, g);
  else
    hipLaunchKernelGGL(gemm_tn_grouped_kernel, dim3(blk), dim3(512), 2 * TD_STAGE, stream, g);
  rc = lc2is_check_launch();
  if (rc || red == 0) return rc;
  hipLaunchKernelGGL(slab_reduce_grouped_kernel, dim3(red), dim3(256), 0, stream, g);
  return lc2is_check_launch();
}

extern "C" size_t lc2is_colsum_workspace_bytes(int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  return (size_t)colsum_parts(M) * N * sizeof(float);
}

extern "C" int lc2is_colsum_bf16(const void* dY, int ldy, float* db, int M, int N, int accumulate,
                                 void* workspace, size_t workspace_bytes, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!dY || !db) return LC2IS_ERR_NULL;
  if (M <= 0 || N <= 0 || N % 8 || ldy < N || ldy % 8) return LC2IS_ERR_SHAPE;
  if (!workspace || workspace_bytes < lc2is_colsum_workspace_bytes(M, N)) return LC2IS_ERR_WORKSPACE;
  const int parts = colsum_parts(M);
  hipLaunchKernelGGL(colsum_kernel, dim3((N / 8 + 63) / 64, parts), dim3(256), 0, stream, (const bf16_t*)dY,
                     ldy, (float*)workspace, M, N);
  int rc = lc2is_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(partials_reduce_kernel, dim3((N + 31) / 32, 1), dim3(1024), 0, stream,
                     (const float*)workspace, parts, (size_t)N, (size_t)0, N, db, (float*)nullptr, accumulate);
  return lc2is_check_launch();
}
