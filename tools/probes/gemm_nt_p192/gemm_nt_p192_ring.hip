// C[M,N] = epilogue( A[M,K] · W[N,K]^T ) — the persistent 256x192 form of the LDS-DMA NT GEMM (round 3).
// replaces the same reference calls as gemm_nt.hip (hf CLIPAttention q/k/v/out_proj, CLIPMLP fc1/fc2 and their data gradients:
// reference model/encoder.py:29-30 -> hf modeling_clip.py:298-350); it takes the problems whose N is a multiple of 192 —
// every GEMM of the ViT-B/16 tower (N = 768, 2304, 3072).
//
// Why this shape, and what is new against the 256x256 kernels (measured on MI355X, profiles/r03_store_overlap_probe.txt and
// profiles/r03_gemm_shapes_base.txt):
//  * Tile rounds.  M = 32 x 1025 tokens gives 128 full row tiles; with 256-column tiles the five N = 768 GEMMs of a layer are 384
//    tiles = 1.5 rounds of the 256 CUs (the second round half empty, paid in full) and the N = 2304 one 4.5.  192-column tiles
//    give 512 / 1536 / 2048 tiles = exactly 2 / 6 / 8 rounds of tiles 3/4 the size: 1.5 / 4.5 / 6.0 tile-times instead of 2 / 5 / 6.
//  * Epilogue stores off the critical path.  All CUs reach their epilogue together, and a K = 768 tile writes 1 byte per 768 FLOP:
//    the stores of a round run at the chip's HBM write rate (6.5 TB/s: 5-10 us per tile) while the matrix pipe idles.  The probe
//    shows that a wave that has ISSUED its 16-32 KiB of stores goes on issuing MFMAs at full rate while they drain (13.4 us of
//    MFMAs + 5.1 us of stores = 14.6 us when nothing waits in between) — what stops that in a GEMM is only `s_waitcnt vmcnt`:
//    gfx9 retires a wave's loads AND stores in issue order through one counter, so the first wait for a DMA of the next tile also
//    waits for every store issued before it.  Here the waves of a block form two groups of four that swap roles every tile:
//        loader group L(t): issues every LDS-DMA of tile t, is the only one that waits on vmcnt, and at the end of tile t STORES
//                           the whole tile — its own quarter-tiles from its own registers, the other group's through LDS;
//        the other group  : issues no vector memory operation that it ever waits for during tile t (its stores of tile t-1 drain
//                           under tile t's MFMAs) and hands its outputs to L(t) through LDS at the end; it is L(t+1), and by then
//                           its stores are a whole tile old.
//    So no wave ever waits for a store, and the next tile's first K stage is requested by L(t+1) at the START of tile t's epilogue.
//  * 96 accumulator registers per wave (64x96 wave tile) instead of 128: the epilogue needs no spill games.
//
// Layout: 8 waves as 4 (M) x 2 (N); K tiles of 64; a stage is the X tile (256 rows x 128 B) followed by the W tile (192 rows x 128 B)
// with the 16-byte chunk XOR swizzle of gemm_nt.hip applied on the DMA source side; two stages (2 x 56 KiB).  During an epilogue
// stage 0 receives the next tile's first K tile and the rest of the LDS (stage 1 + the spare 48 KiB = 8 x 13 KiB) holds one
// row-major patch per wave: outputs leave as whole row segments (16 B per lane), inputs of the epilogue (saved pre-activation,
// fp32 residual) arrive the same way.  Accumulation order per output element equals the other NT kernels' (bitwise-equal sums).
#include "gemm_nt_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int P_BM = 256, P_BN = 192, P_BK = 64;
constexpr int P_WM = 64, P_WN = 96, P_TM = 4, P_TN = 6;
constexpr int P_BKS = 32;                           // K width of a ring slot (one MFMA K step)
constexpr int P_SLOT = (P_BM + P_BN) * 64;          // 28672: X rows (256 x 64 B) | W rows (192 x 64 B)
constexpr int P_NSLOT = 4;
constexpr int P_STAGE = 2 * P_SLOT;                 // the epilogue patches start behind slots 0 and 1
constexpr int P_PATCH = 13312;                     // 64 rows x 208 B (bf16 rows of 96) or 32 rows x 400 B (fp32 rows of 96), 16-B aligned
constexpr int P_LDS = P_STAGE + 8 * P_PATCH;       // 163840 = all of the CU's LDS
static_assert(P_LDS == 163840 && P_LDS >= P_NSLOT * P_SLOT, "LDS plan");
constexpr int P_PITCH16 = 208, P_PITCH32 = 400;
constexpr int P_OOB = 0x7fffffff;
#ifndef P192_STORE_AUX
#define P192_STORE_AUX 2   // cache policy of the epilogue stores: 2 = nt (streaming; measured 9 % faster over a layer than 0 = write-back)
#endif

// (the LDS-DMA builtin lives in a helper without buffer-resource parameters: see gemm_nt.hip)
// One ring slot = a 32-wide K slice of the X and W tiles.  A piece = 16 rows x 64 B = 1 KiB; loader wave lw (0..3) of the loader
// group brings X pieces 4 lw .. 4 lw + 3 and W pieces 3 lw .. 3 lw + 2.  16-byte chunk c of row r sits at chunk position
// c ^ ((-(r >> 2)) & 3) of the row (applied on the source side: lane l fetches chunk (l & 3) ^ ((-(l >> 4)) & 3) of row l >> 2),
// which makes the ds_read_b128 fragment reads bank-conflict free on 64-byte rows.
__device__ __forceinline__ void p192_dma(const bf16_t* A, unsigned a_bytes, const bf16_t* W, unsigned w_bytes, char* slot, int lw,
                                         int a_v, int w_v, int a_step, int w_step, int kb) {
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A, a_bytes);
  const __amdgpu_buffer_rsrc_t rsW = make_rsrc(W, w_bytes);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(slot + (lw * 4 + j) * 1024), 16, a_v + j * a_step, kb, 0, 0);
#pragma unroll
  for (int j = 0; j < 3; ++j)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(slot + P_BM * 64 + (lw * 3 + j) * 1024), 16, w_v + j * w_step, kb, 0, 0);
}

// "all but the n youngest groups of 7 pieces have landed" (n wave-uniform, 0..2)
__device__ __forceinline__ void p192_wait_groups(int n) {
  if (n >= 2) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else if (n == 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void p192_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void p192_barrier() {   // every LDS access of this wave has completed; then all waves'
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

constexpr bool p192_has_aux_in(int act) {
  return act == LC2IS_ACT_DQUICK_GELU || act == LC2IS_ACT_DRELU || act == LC2IS_ACT_MUL_AUX || act == LC2IS_ACT_DGELU_ERF;
}
constexpr bool p192_has_aux_out(int act) {
  return act == LC2IS_ACT_QUICK_GELU || act == LC2IS_ACT_RELU || act == LC2IS_ACT_QUICK_GELU_GRAD || act == LC2IS_ACT_GELU_ERF;
}

// v = A.W^T + bias of one lane's 4 consecutive columns; z = the saved tensor's values (aux-in activations).
// Returns the main output in v and what the forward activations save for the backward in `aux`.  Same arithmetic, in the
// same order, as gemm_epilogue_lds_act in gemm_nt.hip.
template <int ACT>
__device__ __forceinline__ void p192_act(f32x4_t& v, const float (&z)[4], f32x4_t& aux) {
  if constexpr (ACT == LC2IS_ACT_QUICK_GELU_GRAD) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float sg = sigmoidf_fast(1.702f * v[r]);
      aux[r] = sg * (1.f + 1.702f * v[r] * (1.f - sg));
      v[r] *= sg;
    }
  } else if constexpr (p192_has_aux_out(ACT)) {
    aux = v;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      v[r] = (ACT == LC2IS_ACT_RELU) ? fmaxf(v[r], 0.f)
                                     : (ACT == LC2IS_ACT_GELU_ERF ? gelu_erf(v[r]) : v[r] * sigmoidf_fast(1.702f * v[r]));
  } else if constexpr (p192_has_aux_in(ACT)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if constexpr (ACT == LC2IS_ACT_MUL_AUX) {
        v[r] *= z[r];
      } else if constexpr (ACT == LC2IS_ACT_DGELU_ERF) {
        v[r] *= dgelu_erf(z[r]);
      } else if constexpr (ACT == LC2IS_ACT_DRELU) {
        v[r] = z[r] > 0.f ? v[r] : 0.f;
      } else {
        const float sg = sigmoidf_fast(1.702f * z[r]);
        v[r] *= sg * (1.f + 1.702f * z[r] * (1.f - sg));
      }
    }
  }
}

// row-major walk of a wave tile in 16-byte chunks: instruction q (0..11) of a lane covers chunk 64 q + lane; three
// instructions are exactly 16 bf16 rows (12 chunks each) / 8 fp32 rows (24 chunks each)
template <int CPR> struct P192Walk {
  int row[3], ch[3];
  __device__ __forceinline__ explicit P192Walk(int lane) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int c = s * 64 + lane;
      row[s] = c / CPR;
      ch[s] = c - row[s] * CPR;
    }
  }
};

// ACT: epilogue of the bf16-output form (out_bf16 required; no fp32 output, no residual).  F32: fp32-only output
// out = A.W^T + bias (+ fp32 residual, may alias the output), ACT = NONE.
template <int ACT, bool F32>
__global__ __launch_bounds__(512) void gemm_nt_p192_kernel(GemmNtArgs p, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane0 = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wid >> 2, lw = wid & 3;
  const int wm = wid >> 1, wn = wid & 1;
  const int ntn = p.N / P_BN;
  const unsigned a_bytes = (unsigned)p.M * (unsigned)p.lda * 2u, w_bytes = (unsigned)p.N * (unsigned)p.ldw * 2u;
  const int nks = p.K / P_BKS;                              // ring slots per output tile
  const int a_step = 16 * p.lda * 2, w_step = 16 * p.ldw * 2;
  char* const patch = smem + P_STAGE + wid * P_PATCH;
  char* const ppatch = smem + P_STAGE + (wid ^ 4) * P_PATCH;   // the wave of the other group that shares this wave's (wm & 1, wn)

  // request K slice `ks` of output tile `tile` into ring slot `slot` (per-lane offsets are temporaries of the call)
  auto request = [&](int tile, int ks, int slot, int lane) __attribute__((always_inline)) {
    const int m0 = (tile / ntn) * P_BM, n0 = (tile % ntn) * P_BN;
    const int lrow = lane >> 2, lch = (lane & 3) ^ ((-(lane >> 4)) & 3);
    const int a_v = ((m0 + 64 * lw + lrow) * p.lda + lch * 8) * 2, w_v = ((n0 + 48 * lw + lrow) * p.ldw + lch * 8) * 2;
    p192_dma(p.A, a_bytes, p.W, w_bytes, smem + slot * P_SLOT, lw, a_v, w_v, a_step, w_step, ks * P_BKS * 2);
  };

  int t = blockIdx.x;
  if (t >= ntiles) return;
  if (grp == 0) {   // the first tile's loaders: slices 0 and 1
    const int tile0 = xcd_remap(t, ntiles);
    request(tile0, 0, 0, lane0);
    if (nks > 1) request(tile0, 1, 1, lane0);
  }
  for (int it = 0;; ++it) {
    const bool is_L = grp == (it & 1);   // wave-uniform: this wave's group loads during this tile and stores at its end
    int lane = lane0;
    asm volatile("" : "+v"(lane));        // per-tile lane constants are rebuilt from a laundered id: nothing but the accumulators lives across the epilogue
    const int tile = xcd_remap(t, ntiles);
    const int m0 = (tile / ntn) * P_BM, n0 = (tile % ntn) * P_BN;
    const int frow = lane & 15, g = lane >> 4;
    const int kc_off = (g ^ ((-(frow >> 2)) & 3)) << 4;
    const int x_frag = (wm * P_WM + frow) * 64 + kc_off;
    const int w_frag = P_BM * 64 + (wn * P_WN + frow) * 64 + kc_off;
    // the bias of the wave's columns is requested HERE, a whole tile ahead of its use: hipcc's vmcnt bookkeeping does not count
    // LDS-DMA instructions, so its wait for a bias load issued in the epilogue (behind the next tile's first DMAs) would be
    // a wait for those DMAs
    f32x4_t bvs[P_TN];
#pragma unroll
    for (int i = 0; i < P_TN; ++i) {
      bvs[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if (p.bias) bvs[i] = *(const f32x4_t*)(p.bias + n0 + wn * P_WN + i * 16 + g * 4);
    }
    const int t_next = t + gridDim.x;
    const bool more = t_next < ntiles;
    const int tile_n = more ? xcd_remap(t_next, ntiles) : 0;
    // slices 0 and 1 of the next tile go into slots 0 and 1 during the LAST two K steps when the slice count is a multiple of 4
    // (those steps read slots 2 and 3); otherwise at the start of the epilogue (slots 0 and 1 lie below the patches)
    const bool early = more && (nks & 3) == 0;

    // slice 0 has landed (only the loaders wait on vmcnt: slice 1 may still be in flight); every wave is done with the previous
    // epilogue's patches, which cover slots 2 and 3
    if (is_L) p192_wait_groups(nks > 1 ? 1 : 0);
    p192_barrier();
    f32x4_t acc[P_TN][P_TM];
#pragma unroll
    for (int i = 0; i < P_TN; ++i)
#pragma unroll
      for (int j = 0; j < P_TM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // Ring discipline.  In K step s the loaders request slice s + 3 into slot (s + 3) & 3 = the slot read in step s - 1, which
    // every wave has left at the barrier that closed step s - 1; slice s + 1 was requested two steps ago and is awaited at
    // the end of step s with a COUNTED wait that leaves the two younger groups in flight.  A request therefore has two whole
    // steps to land: its latency (from beyond L2: the K slices of a GEMM are a pure stream) is off the critical path, which with
    // one 64-wide stage in flight it was not (1.5 us per 64 of K against 0.83 us of MFMAs, whatever the tile width).
    for (int s_ = 0; s_ < nks; ++s_) {
      const char* cur = smem + (s_ & 3) * P_SLOT;
      if (is_L) {
        if (s_ == 0 && nks > 2) request(tile, 2, 2, lane);
        if (s_ + 3 < nks) request(tile, s_ + 3, (s_ + 3) & 3, lane);
      } else if (early && s_ + 2 >= nks) {
        request(tile_n, s_ + 2 - nks, s_ + 2 - nks, lane);
      }
      bf16x8_t xf[P_TM], wf[P_TN];
#pragma unroll
      for (int j = 0; j < P_TM; ++j) xf[j] = *(const bf16x8_t*)(cur + x_frag + j * 16 * 64);
#pragma unroll
      for (int i = 0; i < P_TN; ++i) wf[i] = *(const bf16x8_t*)(cur + w_frag + i * 16 * 64);
#pragma unroll
      for (int i = 0; i < P_TN; ++i)
#pragma unroll
        for (int j = 0; j < P_TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
      if (is_L) {   // slice s + 1 has landed; the groups requested after it (at most slices s + 2, s + 3) stay in flight
        const int hi = s_ + 3 < nks - 1 ? s_ + 3 : nks - 1;
        p192_wait_groups(hi - (s_ + 1) > 0 ? hi - (s_ + 1) : 0);
      }
      __builtin_amdgcn_s_barrier();   // ... for every wave; all reads of this step's slot are done
    }

    // ------------------------------------------------------------------ epilogue ----
    int le = lane0;
    asm volatile("" : "+v"(le));
#pragma unroll
    for (int i = 0; i < P_TN; ++i) asm volatile("" : "+v"(bvs[i]));   // (the compiler's wait for the bias loads sits here, ahead of the DMAs below)
    if (!is_L && more && !early) {   // next tile's loaders: every slot is free, the patches start above slots 0 and 1
      request(tile_n, 0, 0, le);
      if (nks > 1) request(tile_n, 1, 1, le);
    }
    const int efrow = le & 15, eg = le >> 4;
    const int mw = m0 + wm * P_WM, nw = n0 + wn * P_WN;           // origin of this wave's tile
    const int mwp = m0 + (wm ^ 2) * P_WM;                          // ... and of the partner's (same columns)

    if constexpr (F32) {
      const P192Walk<24> wk(le);
      const __amdgpu_buffer_rsrc_t rsR = make_rsrc(p.resid, p.resid ? (unsigned)p.M * (unsigned)p.ldr * 4u : 0u);
      const __amdgpu_buffer_rsrc_t rsO = make_rsrc(p.out_f32, (unsigned)p.M * (unsigned)p.ldf * 4u);
      // (1) bias and the fp32 residual of the wave's OWN tile: rows arrive as whole 384-byte segments through the patch, 32 rows
      //     at a time; every load of this wave precedes every store it will issue
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (p.resid) {
          i32x4_t rv[12];
#pragma unroll
          for (int q = 0; q < 12; ++q) {
            const int row = 32 * half + 8 * (q / 3) + wk.row[q % 3], m = mw + row;
            rv[q] = __builtin_amdgcn_raw_buffer_load_b128(rsR, m < p.M ? (m * p.ldr + nw + wk.ch[q % 3] * 4) * 4 : P_OOB, 0, 0);
          }
#pragma unroll
          for (int q = 0; q < 12; ++q)
            *(i32x4_t*)(patch + (8 * (q / 3) + wk.row[q % 3]) * P_PITCH32 + wk.ch[q % 3] * 16) = rv[q];
        }
#pragma unroll
        for (int i = 0; i < P_TN; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            f32x4_t v = acc[i][2 * half + jj] + bvs[i];
            if (p.resid) v += *(const f32x4_t*)(patch + (jj * 16 + efrow) * P_PITCH32 + (i * 16 + eg * 4) * 4);
            acc[i][2 * half + jj] = v;
          }
        p192_lds_fence();   // (the second half's residual rows overwrite the patch)
      }
      // (2) 32 rows at a time: own rows -> patch; the storers flush their own, then the partner's
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (half) p192_barrier();   // the storers are done with the partner patches of the first half
#pragma unroll
        for (int i = 0; i < P_TN; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
            *(f32x4_t*)(patch + (jj * 16 + efrow) * P_PITCH32 + (i * 16 + eg * 4) * 4) = acc[i][2 * half + jj];
        if (is_L) {
#pragma unroll
          for (int q = 0; q < 12; ++q) {
            const int prow = 8 * (q / 3) + wk.row[q % 3], m = mw + 32 * half + prow;
            const i32x4_t v = *(const i32x4_t*)(patch + prow * P_PITCH32 + wk.ch[q % 3] * 16);
            __builtin_amdgcn_raw_buffer_store_b128(v, rsO, m < p.M ? (m * p.ldf + nw + wk.ch[q % 3] * 4) * 4 : P_OOB, 0, P192_STORE_AUX);   // nt
          }
        }
        p192_barrier();             // the other group's rows are in their patches
        if (is_L) {
#pragma unroll
          for (int q = 0; q < 12; ++q) {
            const int prow = 8 * (q / 3) + wk.row[q % 3], m = mwp + 32 * half + prow;
            const i32x4_t v = *(const i32x4_t*)(ppatch + prow * P_PITCH32 + wk.ch[q % 3] * 16);
            __builtin_amdgcn_raw_buffer_store_b128(v, rsO, m < p.M ? (m * p.ldf + nw + wk.ch[q % 3] * 4) * 4 : P_OOB, 0, P192_STORE_AUX);
          }
        }
      }
    } else {
      const P192Walk<12> wk(le);
      const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.aux_in, p.aux_in ? (unsigned)p.M * (unsigned)p.ldx * 2u : 0u);
      const __amdgpu_buffer_rsrc_t rsY = make_rsrc(p.aux_out, p.aux_out ? (unsigned)p.M * (unsigned)p.ldy * 2u : 0u);
      const __amdgpu_buffer_rsrc_t rsO = make_rsrc(p.out_bf16, (unsigned)p.M * (unsigned)p.ldo * 2u);
      // (1) the saved tensor of the backward activations: own tile, whole 192-byte row segments -> patch -> MFMA layout
      if constexpr (p192_has_aux_in(ACT)) {
        i32x4_t zv[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) {
          const int row = 16 * (q / 3) + wk.row[q % 3], m = mw + row;
          zv[q] = __builtin_amdgcn_raw_buffer_load_b128(rsX, m < p.M ? (m * p.ldx + nw + wk.ch[q % 3] * 8) * 2 : P_OOB, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 12; ++q)
          *(i32x4_t*)(patch + (16 * (q / 3) + wk.row[q % 3]) * P_PITCH16 + wk.ch[q % 3] * 16) = zv[q];
      }
      // (2) bias, activation / derivative.  What the forward activations save for the backward (pre-activation / derivative)
      //     goes into the patch right away and leaves first; the main output waits in the accumulators
      constexpr bool two = p192_has_aux_out(ACT);
      const bool aux_pass = two && p.aux_out;   // block-uniform
#pragma unroll
      for (int i = 0; i < P_TN; ++i)
#pragma unroll
        for (int j = 0; j < P_TM; ++j) {
          f32x4_t v = acc[i][j] + bvs[i], aux = f32x4_t{0.f, 0.f, 0.f, 0.f};
          float z[4] = {0.f, 0.f, 0.f, 0.f};
          if constexpr (p192_has_aux_in(ACT)) {
            const i32x2_t zk = *(const i32x2_t*)(patch + (j * 16 + efrow) * P_PITCH16 + (i * 16 + eg * 4) * 2);
            z[0] = bf16_to_f32((bf16_t)(zk[0] & 0xffff)); z[1] = bf16_to_f32((bf16_t)((unsigned)zk[0] >> 16));
            z[2] = bf16_to_f32((bf16_t)(zk[1] & 0xffff)); z[3] = bf16_to_f32((bf16_t)((unsigned)zk[1] >> 16));
          }
          p192_act<ACT>(v, z, aux);
          acc[i][j] = v;
          if constexpr (two) {
            if (aux_pass) {
              const i32x2_t pk = {(int)pack_bf16x2(aux[0], aux[1]), (int)pack_bf16x2(aux[2], aux[3])};
              *(i32x2_t*)(patch + (j * 16 + efrow) * P_PITCH16 + (i * 16 + eg * 4) * 2) = pk;
            }
          }
        }
      if constexpr (p192_has_aux_in(ACT)) p192_lds_fence();   // the patch is about to be overwritten with outputs
      // (3) one tensor at a time through the patches: the storers flush their own rows, barrier, the partner's rows
      auto flush = [&](const __amdgpu_buffer_rsrc_t rsT, int ldt) __attribute__((always_inline)) {
        if (is_L) {
#pragma unroll
          for (int q = 0; q < 12; ++q) {
            const int row = 16 * (q / 3) + wk.row[q % 3], m = mw + row;
            const i32x4_t v = *(const i32x4_t*)(patch + row * P_PITCH16 + wk.ch[q % 3] * 16);
            __builtin_amdgcn_raw_buffer_store_b128(v, rsT, m < p.M ? (m * ldt + nw + wk.ch[q % 3] * 8) * 2 : P_OOB, 0, P192_STORE_AUX);   // nt
          }
        }
        p192_barrier();   // the other group's rows are in their patches
        if (is_L) {
#pragma unroll
          for (int q = 0; q < 12; ++q) {
            const int row = 16 * (q / 3) + wk.row[q % 3], m = mwp + row;
            const i32x4_t v = *(const i32x4_t*)(ppatch + row * P_PITCH16 + wk.ch[q % 3] * 16);
            __builtin_amdgcn_raw_buffer_store_b128(v, rsT, m < p.M ? (m * ldt + nw + wk.ch[q % 3] * 8) * 2 : P_OOB, 0, P192_STORE_AUX);
          }
        }
      };
      if constexpr (two) {
        if (aux_pass) {
          flush(rsY, p.ldy);
          p192_barrier();   // the storers are done with the partner patches of the saved tensor
        }
      }
#pragma unroll
      for (int i = 0; i < P_TN; ++i)
#pragma unroll
        for (int j = 0; j < P_TM; ++j) {
          const f32x4_t v = acc[i][j];
          const i32x2_t pk = {(int)pack_bf16x2(v[0], v[1]), (int)pack_bf16x2(v[2], v[3])};
          *(i32x2_t*)(patch + (j * 16 + efrow) * P_PITCH16 + (i * 16 + eg * 4) * 2) = pk;
        }
      flush(rsO, p.ldo);
    }
    if (!more) break;
    t = t_next;
  }
}

template <int ACT, bool F32>
int p192_launch_t(const GemmNtArgs& a, hipStream_t stream) {
  auto kern = gemm_nt_p192_kernel<ACT, F32>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS) != hipSuccess) return LC2IS_ERR_LAUNCH;
    attr_set = true;
  }
  const int ntiles = ((a.M + P_BM - 1) / P_BM) * (a.N / P_BN);
  const int grid = ntiles < 256 ? ntiles : 256;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), P_LDS, stream, a, ntiles);
  return lc2is_check_launch();
}

}  // namespace

bool lc2is_p192_ok(const void* args) {
  const GemmNtArgs& a = *(const GemmNtArgs*)args;
  const double lim = 2147483648.0;
  if (a.N % P_BN || a.K % P_BK || a.M < P_BM) return false;
  if ((double)(a.M + 256) * a.lda * 2.0 >= lim || (double)(a.N + 256) * a.ldw * 2.0 >= lim) return false;
  const bool f32 = a.out_f32 && !a.out_bf16 && !a.aux_out && !a.aux_in && a.act == LC2IS_ACT_NONE;
  if (f32) return a.ldf % 4 == 0 && (!a.resid || a.ldr % 4 == 0) && (double)a.M * a.ldf * 4.0 < lim && (!a.resid || (double)a.M * a.ldr * 4.0 < lim);
  if (!a.out_bf16 || a.out_f32 || a.resid) return false;
  if (a.ldo % 8 || (a.aux_out && a.ldy % 8) || (a.aux_in && a.ldx % 8)) return false;
  if (p192_has_aux_in(a.act) && !a.aux_in) return false;
  if (a.act == LC2IS_ACT_GELU_ERF) return false;   // (libm erff in this epilogue spills; the 256x256 / 128x128 kernels keep it)
  return (double)a.M * a.ldo * 2.0 < lim && (!a.aux_out || (double)a.M * a.ldy * 2.0 < lim) && (!a.aux_in || (double)a.M * a.ldx * 2.0 < lim);
}

int lc2is_launch_p192(const void* args, hipStream_t stream) {
  if (!lc2is_p192_ok(args)) return LC2IS_ERR_UNSUPPORTED;
  const GemmNtArgs& a = *(const GemmNtArgs*)args;
  if (a.out_f32) return p192_launch_t<LC2IS_ACT_NONE, true>(a, stream);
  switch (a.act) {
    case LC2IS_ACT_NONE: return p192_launch_t<LC2IS_ACT_NONE, false>(a, stream);
    case LC2IS_ACT_QUICK_GELU: return p192_launch_t<LC2IS_ACT_QUICK_GELU, false>(a, stream);
    case LC2IS_ACT_DQUICK_GELU: return p192_launch_t<LC2IS_ACT_DQUICK_GELU, false>(a, stream);
    case LC2IS_ACT_RELU: return p192_launch_t<LC2IS_ACT_RELU, false>(a, stream);
    case LC2IS_ACT_DRELU: return p192_launch_t<LC2IS_ACT_DRELU, false>(a, stream);
    case LC2IS_ACT_DGELU_ERF: return p192_launch_t<LC2IS_ACT_DGELU_ERF, false>(a, stream);
    case LC2IS_ACT_QUICK_GELU_GRAD: return p192_launch_t<LC2IS_ACT_QUICK_GELU_GRAD, false>(a, stream);
    case LC2IS_ACT_MUL_AUX: return p192_launch_t<LC2IS_ACT_MUL_AUX, false>(a, stream);
    default: return LC2IS_ERR_UNSUPPORTED;
  }
}
