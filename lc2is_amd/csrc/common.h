// Shared device helpers for the lc2is_amd gfx950 (CDNA4 / MI355X) kernels.
// Everything here is written for wave64 + MFMA + LDS; there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

typedef unsigned short bf16_t;  // raw bf16 bits in HBM

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
typedef __attribute__((ext_vector_type(2))) int i32x2_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

#define LC2IS_OK 0
#define LC2IS_ERR_SHAPE (-1)
#define LC2IS_ERR_NULL (-2)
#define LC2IS_ERR_UNSUPPORTED (-3)
#define LC2IS_ERR_WORKSPACE (-4)
#define LC2IS_ERR_LAUNCH (-5)

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __builtin_bit_cast(float, ((unsigned)v) << 16);
}

// round-to-nearest-even; NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}

// two floats -> one dword of two bf16 (lo in bits 0..15).  As a 2-vector conversion hipcc emits ONE v_cvt_pk_bf16_f32; the scalar
// form `f32_to_bf16(lo) | f32_to_bf16(hi) << 16` it compiled to two of them (one useful half each) plus a v_or_b32_sdwa — three
// vector instructions per pair in every bf16 epilogue (round 5: 396 -> 132 of the ~1300 in the fc1 epilogue).  Same rounding
// (nearest-even, NaN stays NaN): bitwise the same outputs.
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
#if defined(LC2IS_OLD_PACK)   // (A/B builds only)
  return (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
#endif
  typedef float f32x2_v __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));
  const f32x2_v v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_v));
}

// 128-bit buffer resource over [base, base+bytes): out-of-range loads return 0,
// out-of-range stores are dropped (hardware range check) — used for every tile edge.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// LDS-DMA (buffer_load_dwordx4 ... lds: 64 lanes x 16 bytes land at LDS address `lds_addr` + 16 * lane, no VGPR destination) issued
// by inline asm.  Why not __builtin_amdgcn_raw_ptr_buffer_load_lds: hipcc's waitcnt insertion treats the builtin as a pending LDS
// WRITE and puts `s_waitcnt vmcnt(..)` in front of every later LDS read it cannot prove disjoint from it — vmcnt(0) before each
// ds_read_b64_tr_b16 group, vmcnt(6..7) before MFMAs fed by ds_read_b128 — so a DMA ring requested "two tiles ahead" was drained a
// few instructions after every request (found in round 4 in the ISA of all three attention kernels: the tile's HBM / L2 latency sat
// exposed in every tile).  Through asm the compiler does not see the LDS write; every consumer already waits by hand
// (`s_waitcnt vmcnt(N)` + barrier before the first read of a landed tile).  The compiler's own counted waits for ordinary loads
// stay correct: it does not count these DMAs, so it can only wait for MORE than it needs.
// lds_addr must be wave-uniform (it goes to M0, saved and restored around the request); `s_nop 3` + the two s_mov cover the
// VALU-written-SGPR -> VMEM hazard of operands that come fresh from v_readfirstlane.
__device__ __forceinline__ void lds_dma16(const __amdgpu_buffer_rsrc_t rsrc, unsigned lds_addr, int voffset, int soffset) {
  unsigned keep;
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);   // (uniform by contract; this makes it provable)
  soffset = __builtin_amdgcn_readfirstlane(soffset);
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_nop 3\n\t"
      "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds_addr), "v"(voffset), "s"(rsrc), "s"(soffset)
      : "memory");
}

// Every outstanding vector-memory operation of this wave has completed (inline-asm LDS-DMAs included).  The asm statement is
// the ordering point for the compiler's memory operations; the builtin repeats the wait in a form hipcc's waitcnt pass can see, so
// that it stops tracking its own earlier loads as pending — otherwise it re-waits for them with small counts (vmcnt(3..0)) inside
// the loops, which drains the DMA pieces requested since.
__device__ __forceinline__ void wait_vm0() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
}

// Wave64 reductions as DPP row ops (quad_perm x2, row_half_mirror, row_mirror), row_bcast:15 / row_bcast:31 into the
// upper rows and a v_readlane of lane 63: 6 VALU-rate ops and a uniform (SGPR) result.  The __shfl_xor butterfly
// lowers to six ds_bpermute_b32 (an LDS-pipe round trip each) on gfx950 - tools/probes/wave_reduce.hip.
// All 64 lanes must be active at the call.
template <int CTRL, int ROWMASK = 0xf> __device__ __forceinline__ float dpp_or(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                CTRL, ROWMASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_or<0xB1>(0.f, v);
  v += dpp_or<0x4E>(0.f, v);
  v += dpp_or<0x141>(0.f, v);
  v += dpp_or<0x140>(0.f, v);
  v += dpp_or<0x142, 0xa>(0.f, v);
  v += dpp_or<0x143, 0xc>(0.f, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_or<0xB1>(v, v));
  v = fmaxf(v, dpp_or<0x4E>(v, v));
  v = fmaxf(v, dpp_or<0x141>(v, v));
  v = fmaxf(v, dpp_or<0x140>(v, v));
  v = fmaxf(v, dpp_or<0x142, 0xa>(v, v));
  v = fmaxf(v, dpp_or<0x143, 0xc>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---- counter-based dropout RNG ------------------------------------------------------------------------------------
// keep(seed, r, c) is a pure function of a 64-bit seed and a (row, column) coordinate: the forward and the backward
// kernels of a dropout site evaluate it again instead of storing a mask (nn.Dropout / F.dropout / the attention-probability
// dropout of torch's multi_head_attention_forward and hf SwinDropPath are re-stated this way; the stream differs from
// torch's Philox stream, the distribution — Bernoulli(1 - p), survivors scaled by 1/(1 - p) — is the same).
// mix32 = "lowbias32" (two multiplies, three xor-shifts; full avalanche).  The row half is hoisted wherever the row is
// fixed per lane.  16 bits decide: keep iff bits >= thr, thr = round(p * 65536).
struct DropCfg {
  unsigned seed_lo, seed_hi, thr;   // thr == 0: dropout off
  float inv_keep;                   // 1 / (1 - p)
};
__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned drop_row_hash(const DropCfg& d, unsigned r) { return mix32(r + d.seed_lo); }
__device__ __forceinline__ bool drop_keep(const DropCfg& d, unsigned row_hash, unsigned c) {
  return (mix32(row_hash ^ (c + d.seed_hi)) & 0xffffU) >= d.thr;
}
static inline DropCfg make_drop_cfg(float p, unsigned long long seed) {
  DropCfg d;
  d.seed_lo = (unsigned)(seed & 0xffffffffULL);
  d.seed_hi = (unsigned)(seed >> 32);
  long t = (long)(p * 65536.0 + 0.5);
  d.thr = p <= 0.f ? 0u : (unsigned)(t > 65535 ? 65535 : t);
  d.inv_keep = d.thr ? 65536.0f / (float)(65536u - d.thr) : 1.0f;
  return d;
}

// Bijective XCD-aware remap of a 1-D block id: blocks b and b+8 share an XCD
// (round-robin dispatch), so give every XCD one contiguous chunk of the tile list
// and neighbouring tiles (which share operand panels) hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7;
  const int q = nwg >> 3, r = nwg & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// The same idea for grids that run as ROUNDS of 256 co-resident blocks (one block per CU) whose order matters (the grouped
// weight-gradient grid: long blocks first, a finely split tail): inside every full round of 256, XCD x takes the round's 32
// consecutive logical blocks 32x .. 32x+31, so the set of blocks of a round is unchanged; the ragged last round is chunked.
__device__ __forceinline__ int xcd_round_remap(int bid, int nwg) {
  const int full = nwg & ~255;
  if (bid < full) return (bid & ~255) + ((bid & 7) << 5) + ((bid & 255) >> 3);
  return full + xcd_remap(bid - full, nwg - full);
}

namespace {
// out[c] (+)= sum_{k < nparts} ws[k*stride + c] for c < W (W, stride, ws_off_y multiples of 4; ws 16-byte aligned).
// These launches sit between the big kernels of the backward and are latency-, not bandwidth-bound: block = 32 columns
// (8 float4 groups) x 128 row lanes, so a thread has at most nparts/128 independent 16-byte loads in flight (8 for the
// 1024 LayerNorm partials), then a fixed-order 128 -> 32 -> 1 LDS tree (reproducible).  grid.x = ceil(W / 32); grid.y
// selects an (ws, out) pair offset by (y*ws_off_y, out1 if y==1).
__device__ __forceinline__ void partials_reduce_body(const float* __restrict__ w, int nparts, size_t stride, int W,
                                                     float* out, int accumulate, int bx) {
  __shared__ f32x4_t red[128][8];
  __shared__ f32x4_t red2[32][8];
  const int cx = threadIdx.x & 7, ry = threadIdx.x >> 3;
  const int c = bx * 32 + cx * 4;
  f32x4_t s = {0.f, 0.f, 0.f, 0.f};
  if (c < W) {
#pragma unroll 8
    for (int k = ry; k < nparts; k += 128) s += *(const f32x4_t*)(w + (size_t)k * stride + c);
  }
  red[ry][cx] = s;
  __syncthreads();
  if (threadIdx.x < 256) {
    const int r2 = threadIdx.x >> 3;
    red2[r2][cx] = (red[4 * r2][cx] + red[4 * r2 + 1][cx]) + (red[4 * r2 + 2][cx] + red[4 * r2 + 3][cx]);
  }
  __syncthreads();
  if (threadIdx.x < 8 && c < W) {
    f32x4_t t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 32; ++i) t += red2[i][cx];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c + e < W) out[c + e] = accumulate ? out[c + e] + t[e] : t[e];
  }
}

__global__ __launch_bounds__(1024) void partials_reduce_kernel(const float* __restrict__ ws, int nparts,
                                                                size_t stride, size_t ws_off_y, int W, float* out0,
                                                                float* out1, int accumulate) {
  float* out = blockIdx.y ? out1 : out0;
  if (!out) return;
  partials_reduce_body(ws + blockIdx.y * ws_off_y, nparts, stride, W, out, accumulate, blockIdx.x);
}
}  // namespace

// "done once PER DEVICE": hipFuncSetAttribute(MaxDynamicSharedMemorySize) belongs to a function ON a device, so a process that
// drives a second GPU must set it there too (a per-process flag let such a launch fail with ~78 KB of dynamic LDS requested)
// Compute units the tile planners may count on (lc2is_set_cu_budget; 0 = all 256): every large-tile kernel takes a whole CU per block,
// so a CU held by another queue's kernel (RCCL's channels under data parallelism) turns "exactly one round of tiles" into two.
extern "C" int lc2is_get_cu_budget(void);
static inline int lc2is_ncu() { const int b = lc2is_get_cu_budget(); return b > 0 && b < 256 ? b : 256; }
static inline long lc2is_rounds(long blocks) { const int n = lc2is_ncu(); return (blocks + n - 1) / n; }

static inline int lc2is_cur_dev() {
  int d = 0;
  (void)hipGetDevice(&d);
  return d & 63;
}
struct DevOnce {
  std::atomic<unsigned long long> mask{0};
  bool need() const { return !((mask.load(std::memory_order_acquire) >> lc2is_cur_dev()) & 1ull); }
  void done() { mask.fetch_or(1ull << lc2is_cur_dev(), std::memory_order_release); }
};

static inline int lc2is_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? LC2IS_OK : LC2IS_ERR_LAUNCH;
}
