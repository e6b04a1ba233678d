// C[M,N] = epilogue( A[M,K] · W[N,K]^T )   — bf16 operands, fp32 accumulate, gfx950 MFMA.
//
// This one kernel carries every nn.Linear on the LC2IS hot path (forward AND dgrad, since the
// host keeps a transposed bf16 shadow of every weight, so dX = dY · (W^T)^T is again an NT product):
//   hf CLIPAttention q/k/v/out_proj, CLIPMLP fc1/fc2   (reference model/encoder.py:29-30,98-99)
//   torch MultiheadAttention in/out projections, linear1/linear2 of DecoderLayer (model/decoder.py:9-21)
//   TextToPatch.textual / .visual (model/text_patch.py:14-19), prototype logits (model/model.py:50)
//   patch-embedding conv (stride == kernel, so it is a GEMM over im2col'ed patches).
//
// Design (MI355X): BMxBNx64 tile, one 64x64 (or 64x32) sub-tile per wave as 16x16x32 bf16 MFMAs.
// The MFMA "A" operand is the W tile and the "B" operand the X tile, so a lane's 4 accumulator
// registers are 4 CONSECUTIVE output columns n of one row m: the epilogue reads bias / residual /
// saved pre-activation and writes outputs as 8-byte (bf16) or 16-byte (fp32) vectors.
// Tiles are staged global -> VGPR (range-checked buffer loads: rows past M or N read as zero, no
// branches) -> LDS with a 16-byte-chunk XOR swizzle (chunk ^= row & 7) that makes the
// ds_read_b128 fragment reads bank-conflict free; the loads of tile t+1 are issued before the MFMAs of
// tile t and written to the other LDS buffer after them (one barrier per K step).
// Block ids are remapped so that each XCD (private L2) owns a contiguous run of tiles.
#include "gemm_nt_common.h"
#include <cstdlib>

namespace {

// The activation code is a template parameter of both epilogues: with a run-time `act` inside the unrolled sub-tile
// loops the kernels carried every variant inline (25k instructions for the 256x256 kernel, more than the instruction
// cache), and each tile's epilogue was paced by instruction fetch.  One compact straight-line variant runs per launch.
template <int act, int TM, int TN, int WM, int WN>
__device__ __forceinline__ void gemm_epilogue_act(const GemmNtArgs& p, f32x4_t (&acc)[TN][TM], int m0, int n0, int wm,
                                                  int wn, int lane) {
  const int frow = lane & 15, g = lane >> 4;
  // ---- epilogue: lane holds C[m][n..n+3], n = 4*(lane>>4) within the 16-wide sub-tile ----
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = n0 + wn * WN + i * 16 + g * 4;
    if (n >= p.N) continue;
    f32x4_t bv = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (p.bias) bv = *(const f32x4_t*)(p.bias + n);
    // every load of this column of sub-tiles is issued before its first store: the outputs may alias the inputs
    // (in-place residual), so the compiler will not hoist them itself and each sub-tile would pay a full load latency
    constexpr bool kAux = act == LC2IS_ACT_DQUICK_GELU || act == LC2IS_ACT_DRELU || act == LC2IS_ACT_MUL_AUX ||
                          act == LC2IS_ACT_DGELU_ERF || act == LC2IS_ACT_ADD_AUX;
    constexpr int PF = TM < 4 ? TM : 4;   // sub-tiles prefetched together (more would spill beside 128 accumulators)
#pragma unroll
    for (int j0 = 0; j0 < TM; j0 += PF) {
    f32x4_t rv[PF];
    i32x2_t zv[PF];
#pragma unroll
    for (int jj = 0; jj < PF; ++jj) {
      const int m = m0 + wm * WM + (j0 + jj) * 16 + frow;
      rv[jj] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      zv[jj] = i32x2_t{0, 0};
      if (m < p.M) {
        if (p.resid) rv[jj] = *(const f32x4_t*)(p.resid + (size_t)m * p.ldr + n);
        if (kAux) zv[jj] = *(const i32x2_t*)(p.aux_in + (size_t)m * p.ldx + n);
      }
    }
#pragma unroll
    for (int jj = 0; jj < PF; ++jj) {
      const int j = j0 + jj;
      const int m = m0 + wm * WM + j * 16 + frow;
      if (m >= p.M) continue;
      f32x4_t v = acc[i][j] + bv;
      if (act == LC2IS_ACT_QUICK_GELU_GRAD) {
        f32x4_t d;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sg = quick_sigmoid(v[r]);
          d[r] = sg * (1.f + 1.702f * v[r] * (1.f - sg));
          v[r] *= sg;
        }
        if (p.aux_out) {
          i32x2_t pk = {(int)pack_bf16x2(d[0], d[1]), (int)pack_bf16x2(d[2], d[3])};
          *(i32x2_t*)(p.aux_out + (size_t)m * p.ldy + n) = pk;
        }
      } else if (act == LC2IS_ACT_QUICK_GELU || act == LC2IS_ACT_RELU || act == LC2IS_ACT_GELU_ERF) {
        if (p.aux_out) {
          i32x2_t pk = {(int)pack_bf16x2(v[0], v[1]), (int)pack_bf16x2(v[2], v[3])};
          *(i32x2_t*)(p.aux_out + (size_t)m * p.ldy + n) = pk;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
          v[r] = (act == LC2IS_ACT_RELU) ? fmaxf(v[r], 0.f)
                                         : (act == LC2IS_ACT_GELU_ERF ? gelu_erf(v[r]) : v[r] * quick_sigmoid(v[r]));
      } else if (kAux) {
        const i32x2_t zk = zv[jj];
        float z[4] = {bf16_to_f32((bf16_t)(zk[0] & 0xffff)), bf16_to_f32((bf16_t)((unsigned)zk[0] >> 16)),
                      bf16_to_f32((bf16_t)(zk[1] & 0xffff)), bf16_to_f32((bf16_t)((unsigned)zk[1] >> 16))};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (act == LC2IS_ACT_MUL_AUX) {
            v[r] *= z[r];
          } else if (act == LC2IS_ACT_ADD_AUX) {
            v[r] += z[r];
          } else if (act == LC2IS_ACT_DRELU) {
            v[r] = z[r] > 0.f ? v[r] : 0.f;
          } else if (act == LC2IS_ACT_DGELU_ERF) {
            v[r] *= dgelu_erf(z[r]);
          } else {
            const float s = quick_sigmoid(z[r]);
            v[r] *= s * (1.f + 1.702f * z[r] * (1.f - s));
          }
        }
      }
      v += rv[jj];
      if (p.out_f32) *(f32x4_t*)(p.out_f32 + (size_t)m * p.ldf + n) = v;
      if (p.out_bf16) {
        i32x2_t pk = {(int)pack_bf16x2(v[0], v[1]), (int)pack_bf16x2(v[2], v[3])};
        *(i32x2_t*)(p.out_bf16 + (size_t)m * p.ldo + n) = pk;
      }
    }
    }
  }
}

#define LC2IS_ACT_SWITCH(CALL)                          \
  switch (p.act) {                                      \
    case LC2IS_ACT_QUICK_GELU: CALL(LC2IS_ACT_QUICK_GELU); break;             \
    case LC2IS_ACT_RELU: CALL(LC2IS_ACT_RELU); break;                         \
    case LC2IS_ACT_DQUICK_GELU: CALL(LC2IS_ACT_DQUICK_GELU); break;           \
    case LC2IS_ACT_DRELU: CALL(LC2IS_ACT_DRELU); break;                       \
    case LC2IS_ACT_QUICK_GELU_GRAD: CALL(LC2IS_ACT_QUICK_GELU_GRAD); break;   \
    case LC2IS_ACT_MUL_AUX: CALL(LC2IS_ACT_MUL_AUX); break;                   \
    case LC2IS_ACT_GELU_ERF: CALL(LC2IS_ACT_GELU_ERF); break;                 \
    case LC2IS_ACT_DGELU_ERF: CALL(LC2IS_ACT_DGELU_ERF); break;               \
    case LC2IS_ACT_ADD_AUX: CALL(LC2IS_ACT_ADD_AUX); break;                   \
    default: CALL(LC2IS_ACT_NONE); break;                                     \
  }

template <int TM, int TN, int WM, int WN>
__device__ __forceinline__ void gemm_epilogue(const GemmNtArgs& p, f32x4_t (&acc)[TN][TM], int m0, int n0, int wm,
                                              int wn, int lane) {
#define LC2IS_EPI_CALL(A) gemm_epilogue_act<A, TM, TN, WM, WN>(p, acc, m0, n0, wm, wn, lane)
  LC2IS_ACT_SWITCH(LC2IS_EPI_CALL)
#undef LC2IS_EPI_CALL
}

// LDS-staged epilogue for the LDS-DMA kernels (WN == 64): every bf16 tensor touched by the epilogue (saved
// pre-activation in/out, bf16 output) moves between HBM and the wave as whole 128-byte row segments (16 B per
// lane, 8 rows per wave instruction) through a private per-wave LDS patch, instead of 8-byte pieces whose
// 32-byte row fragments cost partial-line writes.  fp32 residual / output keep the direct 16-B (64 B per row)
// form.  Only in-order LDS traffic of one wave touches a patch, so no barrier is needed.
// RC (range-checked form, for the persistent kernel's counted s_waitcnt): every bf16 load / store of the epilogue is a buffer
// access whose out-of-range lanes carry an offset beyond num_records (loads give 0, stores are dropped) — no branch around any
// of them, so a wave issues EXACTLY 16 stores per bf16 output tensor and tile, whatever M and N are.
struct NoHook { __device__ void operator()() const {} };
// `loads_issued` (RC form) is called once, right after the LAST global load of the epilogue has returned (the bias rows when
// the activation reads no saved tensor, else the second row group's saved pre-activations): the persistent kernel requests the
// next tile's first K stage there, so that no wait of the epilogue is for something younger than those DMAs.
template <int act, int TM, int TN, int WM, int WN, bool RC = false, typename Hook = NoHook>
__device__ __forceinline__ void gemm_epilogue_lds_act(const GemmNtArgs& p, f32x4_t (&acc)[TN][TM], int m0, int n0,
                                                      int wm, int wn, int lane, int wid, char* smem, Hook loads_issued = Hook()) {
  static_assert(WN == 64 && TM % 4 == 0, "staged epilogue expects 64-column wave tiles");
  constexpr int OOB = 0x7fffffff;
  const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.aux_in, (RC && p.aux_in) ? (unsigned)p.M * (unsigned)p.ldx * 2u : 0u);
  const __amdgpu_buffer_rsrc_t rsY = make_rsrc(p.aux_out, (RC && p.aux_out) ? (unsigned)p.M * (unsigned)p.ldy * 2u : 0u);
  const __amdgpu_buffer_rsrc_t rsO = make_rsrc(p.out_bf16, (RC && p.out_bf16) ? (unsigned)p.M * (unsigned)p.ldo * 2u : 0u);
  constexpr int PITCH = 144;                 // 128 B of data + 16: conflict-free ds_write_b64 / ds_read_b128
  char* patch = smem + wid * (64 * PITCH);
  const int frow = lane & 15, g = lane >> 4;
  const int srow = lane >> 3, sch = lane & 7;  // flush mapping: 8 rows x 8 chunks per wave instruction
  const int nw = n0 + wn * WN;
  const bool col_ok = (nw + sch * 8) < p.N;

  constexpr bool has_aux = act == LC2IS_ACT_DQUICK_GELU || act == LC2IS_ACT_DRELU || act == LC2IS_ACT_MUL_AUX ||
                           act == LC2IS_ACT_DGELU_ERF || act == LC2IS_ACT_ADD_AUX;
  f32x4_t bvs[TN];   // bias of the wave's columns, requested once per tile
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = nw + i * 16 + g * 4;
    bvs[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (p.bias && n < p.N) bvs[i] = *(const f32x4_t*)(p.bias + n);
  }
  if constexpr (RC && !has_aux) loads_issued();
#pragma unroll
  for (int jg = 0; jg < TM / 4; ++jg) {
    const int mrow0 = m0 + wm * WM + jg * 64;
    if (!RC && mrow0 >= p.M) break;  // wave-uniform
    // ---- (A) saved pre-activation / derivative for the backward epilogues: HBM -> patch (coalesced; the group's
    // eight loads are in flight together — requesting both groups up front costs 64 registers and spills) ----
    if (has_aux) {
      i32x4_t auxv[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int m = mrow0 + it * 8 + srow;
        auxv[it] = i32x4_t{0, 0, 0, 0};
        if constexpr (RC)
          auxv[it] = __builtin_amdgcn_raw_buffer_load_b128(rsX, (m < p.M && col_ok) ? (m * p.ldx + nw + sch * 8) * 2 : OOB, 0, 0);
        else if (m < p.M && col_ok) auxv[it] = *(const i32x4_t*)(p.aux_in + (size_t)m * p.ldx + nw + sch * 8);
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) *(i32x4_t*)(patch + (it * 8 + srow) * PITCH + sch * 16) = auxv[it];
      // (behind the patch stores, i.e. once the loads have RETURNED: hipcc's vmcnt bookkeeping does not count LDS-DMA
      //  instructions, so a wait for a load that has DMAs behind it turns into a wait for the DMAs as well)
      if constexpr (RC) { if (jg == TM / 4 - 1) loads_issued(); }
    }
    // ---- (B) bias, derivative / pre-activation ----
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const f32x4_t bv = bvs[i];
#pragma unroll
      for (int jl = 0; jl < 4; ++jl) {
        f32x4_t v = acc[i][jg * 4 + jl] + bv;
        if (act == LC2IS_ACT_QUICK_GELU_GRAD) {   // activation now, its derivative (bf16) into the patch for (C)
          f32x4_t d;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sg = quick_sigmoid(v[r]);
            d[r] = sg * (1.f + 1.702f * v[r] * (1.f - sg));
            v[r] *= sg;
          }
          i32x2_t pk = {(int)pack_bf16x2(d[0], d[1]), (int)pack_bf16x2(d[2], d[3])};
          *(i32x2_t*)(patch + (jl * 16 + frow) * PITCH + (i * 16 + g * 4) * 2) = pk;
          __builtin_amdgcn_sched_barrier(0);   // one sub-tile's sigmoid temporaries at a time (64 of them spill)
        } else if (has_aux) {
          const i32x2_t zk = *(const i32x2_t*)(patch + (jl * 16 + frow) * PITCH + (i * 16 + g * 4) * 2);
          float z[4] = {bf16_to_f32((bf16_t)(zk[0] & 0xffff)), bf16_to_f32((bf16_t)((unsigned)zk[0] >> 16)),
                        bf16_to_f32((bf16_t)(zk[1] & 0xffff)), bf16_to_f32((bf16_t)((unsigned)zk[1] >> 16))};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (act == LC2IS_ACT_MUL_AUX) {
              v[r] *= z[r];
            } else if (act == LC2IS_ACT_ADD_AUX) {
              v[r] += z[r];
            } else if (act == LC2IS_ACT_DGELU_ERF) {
              v[r] *= dgelu_erf(z[r]);
            } else if (act == LC2IS_ACT_DRELU) {
              v[r] = z[r] > 0.f ? v[r] : 0.f;
            } else {
              const float sg = quick_sigmoid(z[r]);
              v[r] *= sg * (1.f + 1.702f * z[r] * (1.f - sg));
            }
          }
        }
        acc[i][jg * 4 + jl] = v;
      }
    }
    // ---- (C) store the pre-activation (bf16), or the derivative (B) left in the patch ----
    if ((act == LC2IS_ACT_QUICK_GELU || act == LC2IS_ACT_RELU || act == LC2IS_ACT_QUICK_GELU_GRAD ||
         act == LC2IS_ACT_GELU_ERF) && p.aux_out) {
      if (act != LC2IS_ACT_QUICK_GELU_GRAD) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int jl = 0; jl < 4; ++jl) {
            const f32x4_t v = acc[i][jg * 4 + jl];
            i32x2_t pk = {(int)pack_bf16x2(v[0], v[1]), (int)pack_bf16x2(v[2], v[3])};
            *(i32x2_t*)(patch + (jl * 16 + frow) * PITCH + (i * 16 + g * 4) * 2) = pk;
          }
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int r = it * 8 + srow, m = mrow0 + r;
        const i32x4_t v = *(const i32x4_t*)(patch + r * PITCH + sch * 16);
        if constexpr (RC)
          __builtin_amdgcn_raw_buffer_store_b128(v, rsY, (m < p.M && col_ok) ? (m * p.ldy + nw + sch * 8) * 2 : OOB, 0, 2);   // nt: read again only in the backward pass
        else if (m < p.M && col_ok) __builtin_nontemporal_store(v, (i32x4_t*)(p.aux_out + (size_t)m * p.ldy + nw + sch * 8));
      }
    }
    // ---- (D) activation, fp32 residual / output (direct), (E) bf16 output through the patch ----
    // residual rows are requested one sub-tile column (4 loads) ahead of their first store: the outputs may alias
    // them (in-place residual), so the compiler will not hoist the loads itself, and more in flight would spill
    constexpr int RPF = 1;
    static_assert(TN % RPF == 0, "staged epilogue walks sub-tile columns in groups of RPF");
#pragma unroll
    for (int ip = 0; ip < TN; ip += RPF) {
      f32x4_t rv[RPF][4];
#pragma unroll
      for (int ii = 0; ii < RPF; ++ii)
#pragma unroll
        for (int jl = 0; jl < 4; ++jl) {
          const int n = nw + (ip + ii) * 16 + g * 4, m = mrow0 + jl * 16 + frow;
          rv[ii][jl] = f32x4_t{0.f, 0.f, 0.f, 0.f};
          if constexpr (!RC) {   // (the RC form is for bf16 outputs without a residual)
            if (p.resid && m < p.M && n < p.N) rv[ii][jl] = *(const f32x4_t*)(p.resid + (size_t)m * p.ldr + n);
          }
        }
#pragma unroll
      for (int ii = 0; ii < RPF; ++ii) {
        const int i = ip + ii;
        const int n = nw + i * 16 + g * 4;
#pragma unroll
        for (int jl = 0; jl < 4; ++jl) {
          const int m = mrow0 + jl * 16 + frow;
          f32x4_t v = acc[i][jg * 4 + jl];
          if (act == LC2IS_ACT_QUICK_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] * quick_sigmoid(v[r]);
          } else if (act == LC2IS_ACT_GELU_ERF) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
          } else if (act == LC2IS_ACT_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          }
          const bool ok = (m < p.M) && (n < p.N);
          v += rv[ii][jl];
          if constexpr (!RC) {
            if (p.out_f32 && ok) __builtin_nontemporal_store(v, (f32x4_t*)(p.out_f32 + (size_t)m * p.ldf + n));
          }
          if (p.out_bf16) {
            i32x2_t pk = {(int)pack_bf16x2(v[0], v[1]), (int)pack_bf16x2(v[2], v[3])};
            *(i32x2_t*)(patch + (jl * 16 + frow) * PITCH + (i * 16 + g * 4) * 2) = pk;
          }
        }
      }
    }
    if (p.out_bf16) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int r = it * 8 + srow, m = mrow0 + r;
        const i32x4_t v = *(const i32x4_t*)(patch + r * PITCH + sch * 16);
        if constexpr (RC)
          __builtin_amdgcn_raw_buffer_store_b128(v, rsO, (m < p.M && col_ok) ? (m * p.ldo + nw + sch * 8) * 2 : OOB, 0, 0);   // write-back: the next kernel reads it (round 4: +1.5 % of the step against nt, profiles/r04_gemm_store_policy_ab.txt)
        else if (m < p.M && col_ok) __builtin_nontemporal_store(v, (i32x4_t*)(p.out_bf16 + (size_t)m * p.ldo + nw + sch * 8));
      }
    }
  }
}

template <int TM, int TN, int WM, int WN>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmNtArgs& p, f32x4_t (&acc)[TN][TM], int m0, int n0, int wm,
                                                  int wn, int lane, int wid, char* smem) {
#define LC2IS_EPI_CALL(A) gemm_epilogue_lds_act<A, TM, TN, WM, WN>(p, acc, m0, n0, wm, wn, lane, wid, smem)
  LC2IS_ACT_SWITCH(LC2IS_EPI_CALL)
#undef LC2IS_EPI_CALL
}

// fp32-only outputs (the residual-stream GEMMs: out = resid + A.W^T + bias) through LDS as well: in the MFMA layout a wave
// store covers 16 rows x 64 B, half a line per row for the fp32 residual read and the fp32 write alike; staged, each wave
// instruction moves 4 rows x 256 B (whole lines).  64 x 64 fp32 per wave at a 272-byte pitch (conflict-free ds_write_b128 /
// ds_read_b128).  A lane reads and writes the same addresses, so an in-place residual stays safe.
template <int TM, int TN, int WM, int WN>
__device__ __forceinline__ void gemm_epilogue_f32_lds(const GemmNtArgs& p, f32x4_t (&acc)[TN][TM], int m0, int n0, int wm,
                                                      int wn, int lane, int wid, char* smem) {
  static_assert(WN == 64 && TM % 4 == 0, "staged epilogue expects 64-column wave tiles");
  constexpr int PITCH = 272;
  char* patch = smem + wid * (64 * PITCH);
  const int frow = lane & 15, g = lane >> 4;
  const int srow = lane >> 4, sch = lane & 15;   // flush mapping: 4 rows x 16 chunks of 16 B per wave instruction
  const int nw = n0 + wn * WN;
  // range-checked buffer accesses (rows >= M and columns >= N fall outside num_records: loads give 0, stores are dropped):
  // no branches around the 32 loads / 32 stores and one VGPR of offset each (the host keeps M * ld * 4 under 2 GiB)
  const __amdgpu_buffer_rsrc_t rsR = make_rsrc(p.resid, p.resid ? (unsigned)p.M * (unsigned)p.ldr * 4u : 0u);
  const __amdgpu_buffer_rsrc_t rsO = make_rsrc(p.out_f32, (unsigned)p.M * (unsigned)p.ldf * 4u);
  const bool col_ok = (nw + sch * 4) < p.N;
  f32x4_t bvs[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = nw + i * 16 + g * 4;
    bvs[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (p.bias && n < p.N) bvs[i] = *(const f32x4_t*)(p.bias + n);
  }
#pragma unroll
  for (int jg = 0; jg < TM / 4; ++jg) {
    const int mrow0 = m0 + wm * WM + jg * 64;
    if (mrow0 >= p.M) break;  // wave-uniform
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int jl = 0; jl < 4; ++jl)
        *(f32x4_t*)(patch + (jl * 16 + frow) * PITCH + (i * 16 + g * 4) * 4) = acc[i][jg * 4 + jl] + bvs[i];
    const int OOB = 0x7fffffff;
    const int r_off = col_ok ? ((mrow0 + srow) * p.ldr + nw + sch * 4) * 4 : OOB;
    const int o_off = col_ok ? ((mrow0 + srow) * p.ldf + nw + sch * 4) * 4 : OOB;
    const int r_step = p.ldr * 16, o_step = p.ldf * 16;   // 4 rows
    i32x4_t rv[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) rv[it] = __builtin_amdgcn_raw_buffer_load_b128(rsR, col_ok ? r_off + it * r_step : OOB, 0, 0);
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int q = half * 8 + it;
        const f32x4_t v = *(const f32x4_t*)(patch + (q * 4 + srow) * PITCH + sch * 16) + __builtin_bit_cast(f32x4_t, rv[it]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), rsO, col_ok ? o_off + q * o_step : OOB, 0, 2);   // nt
        if (half == 0)   // the second half's row goes out as soon as its register is free
          rv[it] = __builtin_amdgcn_raw_buffer_load_b128(rsR, col_ok ? r_off + (q + 8) * r_step : OOB, 0, 0);
      }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_nt_kernel(GemmNtArgs p) {
  if (gridDim.y > 1) {   // strided-batched launch: one independent problem per blockIdx.y
    const long b = blockIdx.y;
    p.A += b * p.bsA;
    p.W += b * p.bsW;
    if (p.out_bf16) p.out_bf16 += b * p.bsOb;
    if (p.out_f32) p.out_f32 += b * p.bsOf;
  }
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int BK = 64;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int A_CH = BM * 8 / NT;  // 16-byte chunks of the X tile per thread
  constexpr int W_CH = BN * 8 / NT;
  constexpr int STAGE = (BM + BN) * 128;  // bytes per LDS stage (128-byte rows)
  static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "tile/threads mismatch");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int ntn = (p.N + BN - 1) / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;

  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, (unsigned)p.M * (unsigned)p.lda * 2u);
  const __amdgpu_buffer_rsrc_t rsW = make_rsrc(p.W, (unsigned)p.N * (unsigned)p.ldw * 2u);

  int a_goff[A_CH], a_lds[A_CH], w_goff[W_CH], w_lds[W_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    const int c = tid + i * NT, row = c >> 3, kc = c & 7;
    a_goff[i] = ((m0 + row) * p.lda + kc * 8) * 2;
    a_lds[i] = row * 128 + ((kc ^ (row & 7)) << 4);
  }
#pragma unroll
  for (int i = 0; i < W_CH; ++i) {
    const int c = tid + i * NT, row = c >> 3, kc = c & 7;
    w_goff[i] = ((n0 + row) * p.ldw + kc * 8) * 2;
    w_lds[i] = BM * 128 + row * 128 + ((kc ^ (row & 7)) << 4);
  }

  i32x4_t ra[A_CH], rw[W_CH];
  f32x4_t acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row = <multiple of 16> + (lane & 15), so row & 7 == lane & 7
  const int frow = lane & 15, g = lane >> 4, sw = lane & 7;
  const int x_frag = (wm * WM + frow) * 128;
  const int w_frag = BM * 128 + (wn * WN + frow) * 128;
  const int kc_off0 = ((0 + g) ^ sw) << 4, kc_off1 = ((4 + g) ^ sw) << 4;

  const int nk = p.K / BK;

#pragma unroll
  for (int i = 0; i < A_CH; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, a_goff[i], 0, 0);
#pragma unroll
  for (int i = 0; i < W_CH; ++i) rw[i] = __builtin_amdgcn_raw_buffer_load_b128(rsW, w_goff[i], 0, 0);
#pragma unroll
  for (int i = 0; i < A_CH; ++i) *(i32x4_t*)(smem + a_lds[i]) = ra[i];
#pragma unroll
  for (int i = 0; i < W_CH; ++i) *(i32x4_t*)(smem + w_lds[i]) = rw[i];
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const char* cur = smem + (kt & 1) * STAGE;
    char* nxt = smem + ((kt + 1) & 1) * STAGE;
    const bool more = (kt + 1) < nk;
    if (more) {
      const int kb = (kt + 1) * BK * 2;
#pragma unroll
      for (int i = 0; i < A_CH; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, a_goff[i] + kb, 0, 0);
#pragma unroll
      for (int i = 0; i < W_CH; ++i) rw[i] = __builtin_amdgcn_raw_buffer_load_b128(rsW, w_goff[i] + kb, 0, 0);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ko = ks ? kc_off1 : kc_off0;
      bf16x8_t xf[TM], wf[TN];
#pragma unroll
      for (int j = 0; j < TM; ++j) xf[j] = *(const bf16x8_t*)(cur + x_frag + j * 16 * 128 + ko);
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = *(const bf16x8_t*)(cur + w_frag + i * 16 * 128 + ko);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
#pragma unroll
      for (int i = 0; i < A_CH; ++i) *(i32x4_t*)(nxt + a_lds[i]) = ra[i];
#pragma unroll
      for (int i = 0; i < W_CH; ++i) *(i32x4_t*)(nxt + w_lds[i]) = rw[i];
    }
    __syncthreads();
  }

  gemm_epilogue<TM, TN, WM, WN>(p, acc, m0, n0, wm, wn, lane);
}

// (the LDS-DMA builtin has to live in a __device__ helper that takes no __amdgpu_buffer_rsrc_t parameter: with the
//  builtin in a __global__ body, or the opaque descriptor type in a template signature, hipcc's host pass drops the kernel stub)
template <int BM, int A_PIECES, int W_PIECES>
__device__ __forceinline__ void dma_stage(const bf16_t* A, unsigned a_bytes, const bf16_t* W, unsigned w_bytes, char* buf,
                                          int wid, const int* a_goff, const int* w_goff, int kb) {
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A, a_bytes);
  const __amdgpu_buffer_rsrc_t rsW = make_rsrc(W, w_bytes);
#pragma unroll
  for (int j = 0; j < A_PIECES; ++j)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(buf + (wid * A_PIECES + j) * 1024), 16, a_goff[j] + kb, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < W_PIECES; ++j)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(buf + BM * 128 + (wid * W_PIECES + j) * 1024), 16,
                                             w_goff[j] + kb, 0, 0, 0);
}

// ---- LDS-DMA variant for large problems ------------------------------------------------------------------
// Same tile algebra, but the K tiles go global -> LDS directly (buffer_load_dwordx4 ... lds, range-checked):
// no staging VGPRs, no ds_write pass.  An LDS-DMA wave instruction writes 64 x 16 B = 8 rows of 128 B
// linearly, so the XOR swizzle is applied on the per-lane SOURCE address (lane l fetches chunk (l&7)^(l>>3) of
// row l>>3) and the fragment reads use the same involution.  The DMA of tile t+1 is issued right after the
// barrier that publishes tile t and flies during the MFMAs of tile t; the barrier's implicit vmcnt(0) retires
// it.  256x256 tile, 8 waves (2x4), 128x64 per wave: 12 ds_read_b128 per 32 MFMAs.
// EPI: -1 = epilogue chosen at run time (p.act, p.staged_epi), 0..8 = that activation only, -2 = fp32-only output through LDS
template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI = -1>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_nt_dma_kernel(GemmNtArgs p) {
  constexpr int NWAVE = WAVES_M * WAVES_N;
  constexpr int BK = 64;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int A_PIECES = BM / 8 / NWAVE;  // 1-KiB pieces (8 rows) of the X tile per wave
  constexpr int W_PIECES = BN / 8 / NWAVE;
  constexpr int STAGE = (BM + BN) * 128;
  static_assert(BM % (8 * NWAVE) == 0 && BN % (8 * NWAVE) == 0, "tile/waves mismatch");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int ntn = (p.N + BN - 1) / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;

  const unsigned a_bytes = (unsigned)p.M * (unsigned)p.lda * 2u, w_bytes = (unsigned)p.N * (unsigned)p.ldw * 2u;

  // per-lane source offsets of this wave's pieces (piece j of wave w = rows 8*(w*PIECES + j) .. +7)
  const int lrow = lane >> 3, lch = (lane & 7) ^ (lane >> 3);
  int a_goff[A_PIECES], w_goff[W_PIECES];
#pragma unroll
  for (int j = 0; j < A_PIECES; ++j)
    a_goff[j] = ((m0 + 8 * (wid * A_PIECES + j) + lrow) * p.lda + lch * 8) * 2;
#pragma unroll
  for (int j = 0; j < W_PIECES; ++j)
    w_goff[j] = ((n0 + 8 * (wid * W_PIECES + j) + lrow) * p.ldw + lch * 8) * 2;


  f32x4_t acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, g = lane >> 4, sw = lane & 7;
  const int x_frag = (wm * WM + frow) * 128;
  const int w_frag = BM * 128 + (wn * WN + frow) * 128;
  const int kc_off0 = ((0 + g) ^ sw) << 4, kc_off1 = ((4 + g) ^ sw) << 4;
  const int nk = p.K / BK;

  dma_stage<BM, A_PIECES, W_PIECES>(p.A, a_bytes, p.W, w_bytes, smem, wid, a_goff, w_goff, 0);
  __syncthreads();  // emits s_waitcnt vmcnt(0): tile 0 has landed for every wave

  for (int kt = 0; kt < nk; ++kt) {
    const char* cur = smem + (kt & 1) * STAGE;
    if (kt + 1 < nk)
      dma_stage<BM, A_PIECES, W_PIECES>(p.A, a_bytes, p.W, w_bytes, smem + ((kt + 1) & 1) * STAGE, wid, a_goff, w_goff,
                                        (kt + 1) * BK * 2);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ko = ks ? kc_off1 : kc_off0;
      bf16x8_t xf[TM], wf[TN];
#pragma unroll
      for (int j = 0; j < TM; ++j) xf[j] = *(const bf16x8_t*)(cur + x_frag + j * 16 * 128 + ko);
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = *(const bf16x8_t*)(cur + w_frag + i * 16 * 128 + ko);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();  // all reads of `cur` done; DMA of the next tile retired (vmcnt(0)) and published
  }
  if constexpr (EPI == -2) {   // its own instantiation: with all three epilogues in one kernel the allocator spills 400 VGPRs
    gemm_epilogue_f32_lds<TM, TN, WM, WN>(p, acc, m0, n0, wm, wn, lane, wid, smem);
    return;
  } else if constexpr (EPI >= 0) {   // one activation per instantiation (the run-time switch over all nine costs scratch)
    if (p.staged_epi)
      gemm_epilogue_lds_act<EPI, TM, TN, WM, WN>(p, acc, m0, n0, wm, wn, lane, wid, smem);
    else
      gemm_epilogue_act<EPI, TM, TN, WM, WN>(p, acc, m0, n0, wm, wn, lane);
    return;
  } else if constexpr (WN == 64 && TM % 4 == 0) {
    if (p.staged_epi) {
      gemm_epilogue_lds<TM, TN, WM, WN>(p, acc, m0, n0, wm, wn, lane, wid, smem);
      return;
    }
  }
  gemm_epilogue<TM, TN, WM, WN>(p, acc, m0, n0, wm, wn, lane);
}

// ---- ragged-row tail (cfg 17, round 4) ------------------------------------------------------------------------------------------
// The <= 64 rows that the exact-round plans peel off (B x 1025 tokens: 32 rows) used to go through the 64x64 register-staged kernel:
// 12 blocks walking K = 3072 in 48 latency-bound steps = 22 us, as much as the 256x384 plan saves.  Here ONE WAVE owns 16 rows x 16
// columns and the whole K: both operands come straight from global memory in the MFMA fragment layout (a lane's 8 consecutive K
// elements are 16 contiguous bytes of a row), PD K chunks of 32 in flight, no LDS, no barrier.  Same MFMA, same operand roles and the
// same sequential order over K as the tile kernels, and their unstaged epilogue: bitwise equal to tile_cfg 4 for every activation.
// rows_fragment: one such fragment (rows m0.., columns n0..) by the calling wave; ACT < 0: the run-time switch over every epilogue.
template <int PD, int ACT>
__device__ __forceinline__ void rows_fragment(const GemmNtArgs& p, int n0, int m0, int lane) {
  const int frow = lane & 15, g = lane >> 4;
  constexpr int OOB = 0x7fffffff;
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, (unsigned)p.M * (unsigned)p.lda * 2u);
  const __amdgpu_buffer_rsrc_t rsW = make_rsrc(p.W, (unsigned)p.N * (unsigned)p.ldw * 2u);
  const bool n_ok = n0 + frow < p.N;
  const int a_off = ((m0 + frow) * p.lda + g * 8) * 2;   // rows >= M fall outside num_records: zeros
  const int w_off = n_ok ? ((n0 + frow) * p.ldw + g * 8) * 2 : OOB;
  const int kbytes = p.K * 2;
  bf16x8_t wa[PD], xa[PD];
#pragma unroll
  for (int i = 0; i < PD; ++i) {
    const int kb = i * 64;
    wa[i] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rsW, kb < kbytes ? w_off + kb : OOB, 0, 0));
    xa[i] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rsA, kb < kbytes ? a_off + kb : OOB, 0, 0));
  }
  f32x4_t acc[1][1] = {{f32x4_t{0.f, 0.f, 0.f, 0.f}}};
  for (int kb0 = 0; kb0 < kbytes; kb0 += PD * 64) {
#pragma unroll
    for (int i = 0; i < PD; ++i) {
      if (kb0 + i * 64 < kbytes)   // (uniform; chunks past K are never accumulated: the sum stays the tile kernels' sum)
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[i], xa[i], acc[0][0], 0, 0, 0);
      const int kb = kb0 + (i + PD) * 64;
      wa[i] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rsW, kb < kbytes ? w_off + kb : OOB, 0, 0));
      xa[i] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rsA, kb < kbytes ? a_off + kb : OOB, 0, 0));
    }
  }
  if constexpr (ACT < 0) gemm_epilogue<1, 1, 16, 16>(p, acc, m0, n0, 0, 0, lane);   // every epilogue of the tile kernels (same arithmetic, unstaged)
  else gemm_epilogue_act<ACT, 1, 1, 16, 16>(p, acc, m0, n0, 0, 0, lane);
}

template <int PD>
__global__ __launch_bounds__(64) void gemm_nt_rows_kernel(GemmNtArgs p) {
  rows_fragment<PD, -1>(p, blockIdx.x * 16, blockIdx.y * 16, threadIdx.x);
}

// ---- 256x384 tile for the N = 768 problems (cfg 16, round 4) -------------------------------------------------------------------
// The residual-stream GEMMs of the ViT-B tower (out-proj, fc2 and the dgrads into the fp32 stream: N = 768, M = B x 1025) are 3
// column tiles of 256: 384 tiles after the ragged-row peel = 1.5 rounds of the 256 CUs.  With 384-wide tiles they are 2 x 128 =
// 256 tiles: EXACTLY one round.  8 waves (2 x 4), 128 x 96 per wave: 192 accumulator registers; the A fragments of a K sub-step
// are read in two halves of four so that the loop fits 256 registers without scratch; (128 + 96) / (128 x 96) LDS bytes per MFMA
// column instead of (128 + 64) / (128 x 64): 22 % fewer fragment reads per FLOP.  Both stages take the CU's whole 160 KiB of LDS;
// the epilogue's patches overlay them.  Same K order as every other tile shape: bitwise equal to tile_cfg 4.
// EPI = -2: fp32-only output (bias + fp32 residual -> fp32), through LDS in whole 384-byte row segments, 32 rows at a time;
// EPI = 0: bf16-only output (bias, no activation: the dgrads that feed LayerNorm backward and the attention backward), 192-byte segments.
// EPI = 1 (round 5): bf16 output = bf16(acc + bias + aux_in), aux_in the bf16 RESIDUAL STREAM (LC2IS_ACT_ADD_AUX: out-proj, fc2):
//   the 32 x 96 residual rows come in as whole 192-byte segments (the flush mapping), cross the wave's patch into the MFMA layout,
//   are added in fp32 and leave through the same patch; the next row group's segments are requested before the current group's math.
template <int EPI>
__global__ __launch_bounds__(512) void gemm_nt_w384_kernel(GemmNtArgs p) {
  constexpr int BM = 256, BN = 384, WAVES_N = 4, NWAVE = 8, BK = 64;
  constexpr int WM = 128, WN = 96, TM = 8, TN = 6;
  constexpr int A_PIECES = BM / 8 / NWAVE, W_PIECES = BN / 8 / NWAVE;   // 4 + 6 one-KiB pieces per wave and stage
  constexpr int STAGE = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int ntn = p.N / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
  const unsigned a_bytes = (unsigned)p.M * (unsigned)p.lda * 2u, w_bytes = (unsigned)p.N * (unsigned)p.ldw * 2u;

  const int lrow = lane >> 3, lch = (lane & 7) ^ (lane >> 3);
  int a_goff[A_PIECES], w_goff[W_PIECES];
#pragma unroll
  for (int j = 0; j < A_PIECES; ++j) a_goff[j] = ((m0 + 8 * (wid * A_PIECES + j) + lrow) * p.lda + lch * 8) * 2;
#pragma unroll
  for (int j = 0; j < W_PIECES; ++j) w_goff[j] = ((n0 + 8 * (wid * W_PIECES + j) + lrow) * p.ldw + lch * 8) * 2;

  f32x4_t acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, g = lane >> 4, sw = lane & 7;
  const int x_frag = (wm * WM + frow) * 128;
  const int w_frag = BM * 128 + (wn * WN + frow) * 128;
  const int kc_off0 = ((0 + g) ^ sw) << 4, kc_off1 = ((4 + g) ^ sw) << 4;
  const int nk = p.K / BK;

  dma_stage<BM, A_PIECES, W_PIECES>(p.A, a_bytes, p.W, w_bytes, smem, wid, a_goff, w_goff, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const char* cur = smem + (kt & 1) * STAGE;
    if (kt + 1 < nk)
      dma_stage<BM, A_PIECES, W_PIECES>(p.A, a_bytes, p.W, w_bytes, smem + ((kt + 1) & 1) * STAGE, wid, a_goff, w_goff,
                                        (kt + 1) * BK * 2);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ko = ks ? kc_off1 : kc_off0;
      bf16x8_t wf[TN];
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = *(const bf16x8_t*)(cur + w_frag + i * 16 * 128 + ko);
#pragma unroll
      for (int jh = 0; jh < TM; jh += 4) {
        bf16x8_t xf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xf[j] = *(const bf16x8_t*)(cur + x_frag + (jh + j) * 16 * 128 + ko);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][jh + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][jh + j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);   // (keeps the second half's fragment reads out of the first half: 256 registers, no scratch)
      }
    }
    __syncthreads();
  }

  f32x4_t bvs[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = n0 + wn * WN + i * 16 + g * 4;
    bvs[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (p.bias && n < p.N) bvs[i] = *(const f32x4_t*)(p.bias + n);
  }
  if constexpr (EPI == 1) {
    // ---- bf16 epilogue with the bf16 residual stream: out = bf16(acc + bias + aux_in), 16 rows x 96 columns of the wave at a time.
    // The residual rows arrive as whole 192-byte segments (flush mapping: 16 rows x 12 sixteen-byte chunks = 3 wave instructions),
    // cross the wave's patch into the MFMA layout, are added in fp32, and the sums leave through the same patch.  The next 16 rows'
    // segments are requested before the current rows' arithmetic (12 registers in flight; a 32-row pipeline spilled).
    constexpr int PITCHB = WN * 2 + 16;   // 208 B
    char* patchb = smem + wid * (16 * PITCHB);
    const int nwb = n0 + wn * WN;
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.out_bf16, (unsigned)p.M * (unsigned)p.ldo * 2u);
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.aux_in, (unsigned)p.M * (unsigned)p.ldx * 2u);
    constexpr int OOBB = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < TN; ++i)   // bias into the accumulators up front: its 24 registers are free for the epilogue
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] += bvs[i];
    int xoff[3], ooff[3], poff[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int e = k * 64 + lane, row = e / 12, ch = e % 12, col = nwb + ch * 8;
      const bool ok = col < p.N;
      xoff[k] = ok ? ((m0 + wm * WM + row) * p.ldx + col) * 2 : OOBB;   // rows >= M fall outside num_records: zeros / dropped
      ooff[k] = ok ? ((m0 + wm * WM + row) * p.ldo + col) * 2 : OOBB;
      poff[k] = row * PITCHB + ch * 16;
    }
    const int xstep = 16 * p.ldx * 2, ostep = 16 * p.ldo * 2;
    i32x4_t rx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) rx[k] = __builtin_amdgcn_raw_buffer_load_b128(rsX, xoff[k], 0, 0);
#pragma unroll
    for (int h = 0; h < TM; ++h) {
      if (m0 + wm * WM + h * 16 >= p.M) break;   // wave-uniform
#pragma unroll
      for (int k = 0; k < 3; ++k) *(i32x4_t*)(patchb + poff[k]) = rx[k];
      if (h + 1 < TM) {
#pragma unroll
        for (int k = 0; k < 3; ++k) rx[k] = __builtin_amdgcn_raw_buffer_load_b128(rsX, xoff[k], (h + 1) * xstep, 0);
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {   // (the lane that reads a patch word is the lane that overwrites it: in-order LDS traffic of one wave)
        char* w = patchb + frow * PITCHB + (i * 16 + g * 4) * 2;
        const i32x2_t zk = *(const i32x2_t*)w;
        f32x4_t v = acc[i][h];
        v[0] += bf16_to_f32((bf16_t)(zk[0] & 0xffff)); v[1] += bf16_to_f32((bf16_t)((unsigned)zk[0] >> 16));
        v[2] += bf16_to_f32((bf16_t)(zk[1] & 0xffff)); v[3] += bf16_to_f32((bf16_t)((unsigned)zk[1] >> 16));
        *(i32x2_t*)w = i32x2_t{(int)pack_bf16x2(v[0], v[1]), (int)pack_bf16x2(v[2], v[3])};
      }
#pragma unroll
      for (int k = 0; k < 3; ++k)
        __builtin_amdgcn_raw_buffer_store_b128(*(const i32x4_t*)(patchb + poff[k]), rsB, ooff[k] + h * ostep, 0, 0);   // write-back: LayerNorm reads the stream next (row step in the vector offset: see EPI = -3)
    }
  } else if constexpr (EPI == 0) {
    // ---- bf16 epilogue: out = bf16(acc + bias), 32 rows x 96 columns of the wave at a time through its LDS patch ----
    constexpr int PITCHB = WN * 2 + 16;   // 208 B
    char* patchb = smem + wid * (32 * PITCHB);
    const int nwb = n0 + wn * WN;
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.out_bf16, (unsigned)p.M * (unsigned)p.ldo * 2u);
    constexpr int OOBB = 0x7fffffff;
    int brow[3], bch[3];   // flush mapping: 16 rows x 12 sixteen-byte chunks = 3 wave instructions
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int e = k * 64 + lane;
      brow[k] = e / 12;
      bch[k] = e % 12;
    }
#pragma unroll
    for (int jg = 0; jg < TM / 2; ++jg) {
      const int mrow0 = m0 + wm * WM + jg * 32;
      if (mrow0 >= p.M) break;   // wave-uniform
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int jl = 0; jl < 2; ++jl) {
          const f32x4_t v = acc[i][jg * 2 + jl] + bvs[i];
          const i32x2_t pk = {(int)pack_bf16x2(v[0], v[1]), (int)pack_bf16x2(v[2], v[3])};
          *(i32x2_t*)(patchb + (jl * 16 + frow) * PITCHB + (i * 16 + g * 4) * 2) = pk;
        }
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int row = (q / 3) * 16 + brow[q % 3], col = nwb + bch[q % 3] * 8;
        const i32x4_t v = *(const i32x4_t*)(patchb + row * PITCHB + bch[q % 3] * 16);
        const int b_off = col < p.N ? ((mrow0 + row) * p.ldo + col) * 2 : OOBB;
        __builtin_amdgcn_raw_buffer_store_b128(v, rsB, b_off, 0, 0);   // write-back: the next kernel reads it
      }
    }
  } else if constexpr (EPI == -3) {
    // ---- fp32 epilogue + the LayerNorm that follows it (round 5): x' = acc + bias + resid leaves as fp32 (the residual stream, as
    // EPI = -2) AND h = LN(x') leaves as bf16 — the next GEMM's operand — without the LayerNorm kernel's pass over x' (read 4, write
    // 2 bytes per element, 30 us per launch at M = 32 800 x 768, 24 launches per step).  A row's 768 columns live in the
    // accumulators of 2 x 4 waves of TWO blocks (column tiles tn = 0, 1 of one row tile: consecutive logical tiles, co-resident —
    // the grid is one round): per row, each block reduces (mean, M2 = sum of squared deviations from ITS mean) over its 384
    // columns — in registers, across the 4 lanes of a row (g), across the 4 column waves through LDS —, publishes the pair as
    // two 8-byte {value, tag} granules (one agent-scope store each: the datum is its own flag, no ordering needed), polls the
    // partner's, and both combine the halves in the fixed order tn = 0, 1 (Chan's update: the two-pass variance of the LayerNorm
    // kernel, no E[x^2] - mean^2 cancellation): bitwise the same (mean, rstd) on both sides, run to run.  The reader zeroes the
    // granules it consumed (written once, read once per launch): the exchange buffer is all zero between launches, so a
    // captured launch replays without a per-launch epoch.  Spins are bounded (a timeout poisons the row with NaN).
    constexpr int PITCH = WN * 4 + 16;   // 400 B
    char* patch = smem + wid * (32 * PITCH);
    float* lnS = (float*)(smem + NWAVE * 32 * PITCH);   // [256][4]: row sums per column wave
    float* lnQ = lnS + 256 * 4;                         // [256][4]: row sums of squared deviations
    float* lnF = lnQ + 256 * 4;                         // [256][2]: mean, rstd of the whole row
    float* lnG = lnF + 256 * 2;                         // [2][384]: gamma, beta of this block's columns
    const int tn = tile % ntn;
    if (tid < 192) {   // gamma / beta of the block's 384 columns -> LDS (read back as 16-byte fragments in the MFMA layout)
      const float* src = tid < 96 ? p.ln_gamma : p.ln_beta;
      const int c4 = tid < 96 ? tid : tid - 96;
      f32x4_t v = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if (src) v = *(const f32x4_t*)(src + n0 + 4 * c4);
      *(f32x4_t*)(lnG + (tid < 96 ? 0 : 384) + 4 * c4) = v;
    }
    const int nw = n0 + wn * WN;
    const __amdgpu_buffer_rsrc_t rsR = make_rsrc(p.resid, p.resid ? (unsigned)p.M * (unsigned)p.ldr * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rsO = make_rsrc(p.out_f32, (unsigned)p.M * (unsigned)p.ldf * 4u);
    // flush mapping of a 32 x 96 fp32 patch: element k * 64 + lane -> (row, 16-byte chunk) of an 8-row block; the 8-row blocks of
    // the wave's 128 rows are uniform steps (scalar offsets), so a lane keeps three offsets per stream (M is a multiple of 256 here)
    int r_off[3], o_off[3], p_off[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int e = k * 64 + lane, row = e / 24, ch = e % 24;
      r_off[k] = ((m0 + wm * WM + row) * p.ldr + nw + ch * 4) * 4;
      o_off[k] = ((m0 + wm * WM + row) * p.ldf + nw + ch * 4) * 4;
      p_off[k] = row * PITCH + ch * 16;
    }
    const int r_step = 8 * p.ldr * 4, o_step = 8 * p.ldf * 4;
    // phase A: x' through the wave's patch (EPI = -2's flush), written back to it and read again in the MFMA layout: acc = x'
#pragma unroll
    for (int i = 0; i < TN; ++i)   // bias into the accumulators up front: its 24 registers are free for the epilogue
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] += bvs[i];
#pragma unroll
    for (int jg = 0; jg < TM / 2; ++jg) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int jl = 0; jl < 2; ++jl)
          *(f32x4_t*)(patch + (jl * 16 + frow) * PITCH + (i * 16 + g * 4) * 4) = acc[i][jg * 2 + jl];
#pragma unroll
      for (int hq = 0; hq < 12; hq += 6) {
        i32x4_t rv[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          const int rb = jg * 4 + (hq + q) / 3;   // 8-row block of the wave's rows
          rv[q] = __builtin_amdgcn_raw_buffer_load_b128(rsR, r_off[q % 3], rb * r_step, 0);   // (no residual: zero records, reads as zeros)
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          const int rb = jg * 4 + (hq + q) / 3;
          f32x4_t* slot = (f32x4_t*)(patch + p_off[q % 3] + ((hq + q) / 3) * 8 * PITCH);
          const f32x4_t v = *slot + __builtin_bit_cast(f32x4_t, rv[q]);
          *slot = v;
          // (row step in the VECTOR offset: with a register in the scalar-offset field hipcc inserts no wait state between a 16-byte
          //  buffer store and a VALU write of its data registers — LLVM models that case as hazard-free — and on gfx950 the first
          //  dword of some lanes was stored as the zero a following v_mov_b32 put there: found with this epilogue, round 5)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), rsO, o_off[q % 3] + rb * o_step, 0, 0);   // write-back (non-temporal measured equal in the step: 1076 vs 1077 img/s)
        }
      }
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int jl = 0; jl < 2; ++jl)
          acc[i][jg * 2 + jl] = *(const f32x4_t*)(patch + (jl * 16 + frow) * PITCH + (i * 16 + g * 4) * 4);
      __builtin_amdgcn_sched_barrier(0);
    }
    // phase B: per row, sum and squared deviations over the block's 384 columns
    __builtin_amdgcn_sched_barrier(0);
    const int rbase = wm * WM + frow;   // the lane's rows in the block: rbase + 16 j
    float mh[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      float sj = 0.f;
#pragma unroll
      for (int i = 0; i < TN; ++i) sj += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
      sj += __shfl_xor(sj, 16, 64);
      sj += __shfl_xor(sj, 32, 64);
      if (g == 0) lnS[(rbase + 16 * j) * 4 + wn] = sj;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const f32x4_t t = *(const f32x4_t*)(lnS + (rbase + 16 * j) * 4);
      mh[j] = ((t[0] + t[1]) + (t[2] + t[3])) * (1.0f / 384.0f);
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      float qj = 0.f;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const f32x4_t d = acc[i][j] - mh[j];
        qj += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
      }
      qj += __shfl_xor(qj, 16, 64);
      qj += __shfl_xor(qj, 32, 64);
      if (g == 0) lnQ[(rbase + 16 * j) * 4 + wn] = qj;
    }
    __syncthreads();
    // phase C: one thread per row combines the block's pair with the partner block's
    if (tid < 256) {
      const int gr = m0 + tid;
      const f32x4_t s4 = *(const f32x4_t*)(lnS + tid * 4), q4 = *(const f32x4_t*)(lnQ + tid * 4);
      const float my_mean = ((s4[0] + s4[1]) + (s4[2] + s4[3])) * (1.0f / 384.0f);
      const float my_m2 = (q4[0] + q4[1]) + (q4[2] + q4[3]);
      float cm = my_mean, cq = my_m2;
      if (gr < p.M) {
        if (ntn == 2) {   // (a diagnostic build without the exchange: 75.3 against 77.1 us for out-proj — it costs 1.8 us)
          constexpr unsigned long long TAG = 0x4C4E0001ull << 32;
          unsigned long long* mine = p.ln_xchg + ((size_t)gr * 2 + tn) * 2;
          unsigned long long* theirs = p.ln_xchg + ((size_t)gr * 2 + (tn ^ 1)) * 2;
          __hip_atomic_store(mine, TAG | (unsigned long long)__builtin_bit_cast(unsigned, my_mean), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(mine + 1, TAG | (unsigned long long)__builtin_bit_cast(unsigned, my_m2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          unsigned long long a = 0, b = 0;
          int spins = 0;
          for (;;) {
            a = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            b = __hip_atomic_load(theirs + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (((a & b) >> 32) == (TAG >> 32) || ++spins > (1 << 22)) break;
            __builtin_amdgcn_s_sleep(2);
          }
          float pm = __builtin_bit_cast(float, (unsigned)a), pq = __builtin_bit_cast(float, (unsigned)b);
          if (((a & b) >> 32) != (TAG >> 32)) pm = pq = __builtin_nanf("");   // the partner never arrived: poison, do not hang
          __hip_atomic_store(theirs, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(theirs + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          // halves in the order tn = 0, 1 on both sides (equal counts): mean = m0 + (m1 - m0) / 2, M2 = M2_0 + M2_1 + (m1 - m0)^2 * 192
          const float h0m = tn == 0 ? my_mean : pm, h1m = tn == 0 ? pm : my_mean;
          const float h0q = tn == 0 ? my_m2 : pq, h1q = tn == 0 ? pq : my_m2;
          const float dlt = h1m - h0m;
          cm = h0m + dlt * 0.5f;
          cq = (h0q + h1q) + dlt * dlt * 192.0f;
        }
        const float rstd = rsqrtf(cq / (float)p.N + p.ln_eps);
        lnF[tid * 2] = cm;
        lnF[tid * 2 + 1] = rstd;
        if (tn == 0) {
          if (p.ln_mean) p.ln_mean[gr] = cm;
          if (p.ln_rstd) p.ln_rstd[gr] = rstd;
        }
      } else {
        lnF[tid * 2] = 0.f;
        lnF[tid * 2 + 1] = 0.f;
      }
    }
    __syncthreads();
    // phase D: h = (x' - mean) * rstd * gamma + beta, bf16, through the wave's patch as whole 192-byte row segments (EPI = 0's flush)
    constexpr int PITCHB = WN * 2 + 16;   // 208 B
    char* patchb = patch;                 // (the wave's own region again: 32 x 208 < 32 x 400)
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.ln_out, (unsigned)p.M * (unsigned)p.ldl * 2u);
    int b_off[3], pb_off[3];   // flush mapping of a 32 x 96 bf16 patch: 16 rows x 12 sixteen-byte chunks = 3 wave instructions
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int e = k * 64 + lane, row = e / 12, ch = e % 12;
      b_off[k] = ((m0 + wm * WM + row) * p.ldl + nw + ch * 8) * 2;
      pb_off[k] = row * PITCHB + ch * 16;
    }
    const int b_step = 16 * p.ldl * 2;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jg = 0; jg < TM / 2; ++jg) {
      float mu[2], rs[2];
#pragma unroll
      for (int jl = 0; jl < 2; ++jl) {
        mu[jl] = lnF[(rbase + 16 * (jg * 2 + jl)) * 2];
        rs[jl] = lnF[(rbase + 16 * (jg * 2 + jl)) * 2 + 1];
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const f32x4_t gm = *(const f32x4_t*)(lnG + wn * WN + i * 16 + g * 4);
        const f32x4_t bt = *(const f32x4_t*)(lnG + 384 + wn * WN + i * 16 + g * 4);
#pragma unroll
        for (int jl = 0; jl < 2; ++jl) {
          const f32x4_t v = (acc[i][jg * 2 + jl] - mu[jl]) * rs[jl] * gm + bt;
          const i32x2_t pk = {(int)pack_bf16x2(v[0], v[1]), (int)pack_bf16x2(v[2], v[3])};
          *(i32x2_t*)(patchb + (jl * 16 + frow) * PITCHB + (i * 16 + g * 4) * 2) = pk;
        }
      }
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const i32x4_t v = *(const i32x4_t*)(patchb + pb_off[q % 3] + (q / 3) * 16 * PITCHB);
        __builtin_amdgcn_raw_buffer_store_b128(v, rsB, b_off[q % 3] + (jg * 2 + q / 3) * b_step, 0, 0);   // write-back: the next GEMM reads it
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
  // ---- epilogue: out = acc + bias (+ fp32 residual), 32 rows x 96 columns of the wave at a time through its LDS patch ----
  constexpr int PITCH = WN * 4 + 16;   // 400 B: 16 lanes of a ds_write_b128 group land on 16 distinct bank quads
  char* patch = smem + wid * (32 * PITCH);
  const int nw = n0 + wn * WN;
  const __amdgpu_buffer_rsrc_t rsR = make_rsrc(p.resid, p.resid ? (unsigned)p.M * (unsigned)p.ldr * 4u : 0u);
  const __amdgpu_buffer_rsrc_t rsO = make_rsrc(p.out_f32, (unsigned)p.M * (unsigned)p.ldf * 4u);
  constexpr int OOB = 0x7fffffff;
  // flush mapping: 8 rows x 24 sixteen-byte chunks = 3 wave instructions; element k * 64 + lane -> (row, chunk)
  int frow8[3], fch[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int e = k * 64 + lane;
    frow8[k] = e / 24;
    fch[k] = e % 24;
  }
#pragma unroll
  for (int jg = 0; jg < TM / 2; ++jg) {
    const int mrow0 = m0 + wm * WM + jg * 32;
    if (mrow0 >= p.M) break;   // wave-uniform
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int jl = 0; jl < 2; ++jl)
        *(f32x4_t*)(patch + (jl * 16 + frow) * PITCH + (i * 16 + g * 4) * 4) = acc[i][jg * 2 + jl] + bvs[i];
#pragma unroll
    for (int hq = 0; hq < 12; hq += 6) {
      i32x4_t rv[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int row = ((hq + q) / 3) * 8 + frow8[q % 3], col = nw + fch[q % 3] * 4;
        rv[q] = __builtin_amdgcn_raw_buffer_load_b128(rsR, col < p.N ? ((mrow0 + row) * p.ldr + col) * 4 : OOB, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int row = ((hq + q) / 3) * 8 + frow8[q % 3], col = nw + fch[q % 3] * 4;
        const f32x4_t v = *(const f32x4_t*)(patch + row * PITCH + fch[q % 3] * 16) + __builtin_bit_cast(f32x4_t, rv[q]);
        const int o_off = col < p.N ? ((mrow0 + row) * p.ldf + col) * 4 : OOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), rsO, o_off, 0, 0);   // write-back: LayerNorm reads the stream next
      }
    }
  }
  }
  // ---- folded ragged rows: fragment jobs wid * gridDim.x + blockIdx.x, ... (wave 0 of every block first) ----
  if (p.tail_rows > 0) {
    GemmNtArgs q = p;
    q.M = p.M + p.tail_rows;
    const int nfn = p.N / 16, njobs = nfn * ((p.tail_rows + 15) / 16);
    for (int job = wid * (int)gridDim.x + (int)blockIdx.x; job < njobs; job += 8 * (int)gridDim.x)
      rows_fragment<16, EPI == 1 ? LC2IS_ACT_ADD_AUX : LC2IS_ACT_NONE>(q, (job % nfn) * 16, p.M + (job / nfn) * 16, lane);
  }
}

// ---- persistent form with the two wave groups in PING-PONG (cfg 15) ------------------------------------------------------------
// Tile algebra and epilogue of cfg 4 (bitwise equal to it).  The seam between two tiles of a block: gfx9 retires loads and stores
// of a wave in order through ONE counter, so with the next tile's first-stage DMAs issued from inside the epilogue and exactly S
// stores behind them (the RC epilogue: 16 per bf16 output tensor, no branch around any of them), `s_waitcnt vmcnt(S)` means "the
// DMAs have landed" while the stores are still in flight (round 2's cfg 13, archived under tools/probes/gemm_nt_persist2/, was the
// lockstep form of this kernel).  bf16 outputs only (no fp32 residual).  The K step is no longer executed by all eight
// waves in lockstep ("everybody issues DMA pieces — no MFMA in flight for ~960 cycles —, everybody runs 64 MFMAs on a matrix pipe
// shared with its SIMD partner, everybody waits at the barrier": profiles/r03_gemm_step_stamps.txt, 3440 cycles per K step for
// 2048 cycles of matrix pipe).  The two groups of four waves (wm = 0 / 1; the partners on a SIMD are waves w and w + 4) run HALF
// A K STEP apart, two barriers per K step: while one group is in its COMPUTE segment (24 fragment reads, 64 MFMAs — alone on
// the matrix pipe), the other is in its LOAD segment (inline-asm LDS-DMA requests, counted wait).  With two K-tile buffers:
//   group 0, load segment t:    request its OWN half of A (rows 0..127: only group 0 reads them) of K tile t+1; await A(t)
//   group 1, load segment t:    request W (all 256 rows) and its own half of A of K tile t+1; await its A(t);
//            compute segment t: at the end await W(t+1) — the barrier that closes the segment publishes it to group 0
// W(t+1) may only be requested once BOTH groups have left K tile t-1 (same buffer), i.e. in group 1's load segment; a group's
// own A half is free one interval earlier.  Waits are counted (`s_waitcnt vmcnt(n)`, n from sequence numbers, so the previous
// epilogue's stores in the in-order queue are waited PAST, not for); nothing waits for vmcnt(0) in the steady state.
__device__ __forceinline__ void wait_vmcnt_le(int n) {   // s_waitcnt vmcnt(<= n): even counts up to 56 (a smaller count only waits longer)
  if (n >= 56) { asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); return; }
  switch (n >> 1) {
#define LC2IS_W(k) case k: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * k) : "memory"); break;
    LC2IS_W(0) LC2IS_W(1) LC2IS_W(2) LC2IS_W(3) LC2IS_W(4) LC2IS_W(5) LC2IS_W(6) LC2IS_W(7) LC2IS_W(8) LC2IS_W(9) LC2IS_W(10)
    LC2IS_W(11) LC2IS_W(12) LC2IS_W(13) LC2IS_W(14) LC2IS_W(15) LC2IS_W(16) LC2IS_W(17) LC2IS_W(18) LC2IS_W(19) LC2IS_W(20)
    LC2IS_W(21) LC2IS_W(22) LC2IS_W(23) LC2IS_W(24) LC2IS_W(25) LC2IS_W(26) LC2IS_W(27)
#undef LC2IS_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

__device__ __forceinline__ bf16x8_t lds_read_b128(unsigned addr) {   // by 32-bit LDS address (base register + immediate offset)
  return *(const __attribute__((address_space(3))) bf16x8_t*)(size_t)addr;
}

template <int ACT>
__global__ __launch_bounds__(512) void gemm_nt_pp_kernel(GemmNtArgs p, int ntiles) {
  constexpr int BM = 256, BN = 256, WAVES_N = 4, BK = 64;
  constexpr int WM = 128, WN = 64, TM = 8, TN = 4;
  constexpr int STAGE = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane0 = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int ntn = (p.N + BN - 1) / BN;
  const unsigned a_bytes = (unsigned)p.M * (unsigned)p.lda * 2u, w_bytes = (unsigned)p.N * (unsigned)p.ldw * 2u;
  const int nk = p.K / BK;
  constexpr bool kAuxOut = ACT == LC2IS_ACT_QUICK_GELU || ACT == LC2IS_ACT_RELU || ACT == LC2IS_ACT_QUICK_GELU_GRAD ||
                           ACT == LC2IS_ACT_GELU_ERF;
  constexpr bool kAuxIn = ACT == LC2IS_ACT_DQUICK_GELU || ACT == LC2IS_ACT_DRELU || ACT == LC2IS_ACT_MUL_AUX || ACT == LC2IS_ACT_DGELU_ERF ||
                          ACT == LC2IS_ACT_ADD_AUX;
  // stores a wave issues BEHIND the next tile's first-stage DMAs (RC epilogue; block-uniform): all 16 per output tensor when the
  // DMAs go out after the bias loads, the second row group's 8 when they go out after that group's saved-tensor loads
  const int nstores = ((p.out_bf16 ? 1 : 0) + ((kAuxOut && p.aux_out) ? 1 : 0)) * (kAuxIn ? 8 : 16);
  const unsigned smem_a = (unsigned)(size_t)LDS_PTR(smem);

  int issued = 0;                    // DMA pieces + counted stores this wave has issued so far
  int seq_a[2] = {0, 0}, seq_w[2] = {0, 0};   // `issued` right after the requests of a K tile's A pieces / W pieces, per buffer
  int a_goff[4], w_goff[8];
  auto tile_offsets = [&](int tt, int lane) {
    const int tile = xcd_remap(tt, ntiles);
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    const int lrow = lane >> 3, lch = (lane & 7) ^ (lane >> 3);
#pragma unroll
    for (int j = 0; j < 4; ++j) a_goff[j] = ((m0 + 8 * (wid * 4 + j) + lrow) * p.lda + lch * 8) * 2;   // pieces 4 wid .. 4 wid + 3: rows of this group's half
#pragma unroll
    for (int j = 0; j < 8; ++j) w_goff[j] = ((n0 + 8 * (wn * 8 + j) + lrow) * p.ldw + lch * 8) * 2;    // (group 1 only) pieces 8 wn .. 8 wn + 7
  };
  auto issue_a = [&](int kt) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.A, a_bytes);
    const unsigned base = smem_a + (kt & 1) * STAGE + wid * 4096;
#pragma unroll
    for (int j = 0; j < 4; ++j) lds_dma16(rs, base + j * 1024, a_goff[j], kt * BK * 2);
    issued += 4;
    seq_a[kt & 1] = issued;
  };
  auto issue_w = [&](int kt) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.W, w_bytes);
    const unsigned base = smem_a + (kt & 1) * STAGE + BM * 128 + wn * 8192;
#pragma unroll
    for (int j = 0; j < 8; ++j) lds_dma16(rs, base + j * 1024, w_goff[j], kt * BK * 2);
    issued += 8;
    seq_w[kt & 1] = issued;
  };

  int t = blockIdx.x;
  if (t >= ntiles) return;
  tile_offsets(t, lane0);
  if (wm == 1) issue_w(0);
  issue_a(0);

  for (;;) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int tile = xcd_remap(t, ntiles);
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    const int frow = lane & 15, g = lane >> 4, sw = lane & 7;
    const unsigned x_frag = smem_a + (wm * WM + frow) * 128;
    const unsigned w_frag = smem_a + BM * 128 + (wn * WN + frow) * 128;
    const int kc_off0 = ((0 + g) ^ sw) << 4, kc_off1 = ((4 + g) ^ sw) << 4;

    wait_vmcnt_le(issued - seq_a[0]);   // this wave's pieces of K tile 0 (A is requested last) — past the previous epilogue's stores
    __builtin_amdgcn_s_barrier();       // K tile 0 is published; every wave has left the previous tile's patches (they overlay buffer 1)
    f32x4_t acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (wm == 1) __builtin_amdgcn_s_barrier();   // the second group runs one barrier (half a K step) behind
    for (int kt = 0; kt < nk; ++kt) {
      const bool more_k = kt + 1 < nk, steady = kt >= 1 && more_k;
      // ---- load segment ----
      if (more_k) {
        if (wm == 1) issue_w(kt + 1);   // (W first: it is awaited first)
        issue_a(kt + 1);
      }
      if (steady) {                     // this group's A(kt) pieces have landed; only the requests just made stay in flight
        if (wm == 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        wait_vmcnt_le(issued - seq_a[kt & 1]);
      }
      __builtin_amdgcn_s_barrier();     // (for group 0 this barrier also publishes W(kt): group 1 awaited it one segment ago)
      // ---- compute segment: alone on the matrix pipe (the SIMD partner is in its load segment) ----
      __builtin_amdgcn_sched_barrier(0);
      const unsigned bo = (kt & 1) * STAGE;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int ko = ks ? kc_off1 : kc_off0;
        bf16x8_t xf[TM], wf[TN];
#pragma unroll
        for (int j = 0; j < TM; ++j) xf[j] = lds_read_b128(x_frag + bo + j * 2048 + ko);
#pragma unroll
        for (int i = 0; i < TN; ++i) wf[i] = lds_read_b128(w_frag + bo + i * 2048 + ko);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (wm == 1 && more_k) {          // W(kt+1) has landed (its A pieces may still fly): the barrier publishes it to group 0
        if (kt >= 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else wait_vmcnt_le(issued - seq_w[(kt + 1) & 1]);
      }
      __builtin_amdgcn_s_barrier();
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();   // group 1's last compute segment: the groups meet again

    t += gridDim.x;
    const bool more = t < ntiles;
    int lane_e = lane0;
    asm volatile("" : "+v"(lane_e));
    // buffer 0 is free (every wave is past its last fragment read): the next tile's K tile 0 is requested from inside the
    // epilogue, behind its last global load; the patches overlay buffer 1 (+ 9 KiB)
    gemm_epilogue_lds_act<ACT, TM, TN, WM, WN, true>(p, acc, m0, n0, wm, wn, lane_e, wid, smem + STAGE, [&]() {
      if (more) {
        tile_offsets(t, lane_e);
        if (wm == 1) issue_w(0);
        issue_a(0);
      }
    });
    issued += nstores;
    if (!more) break;
  }
  // ---- folded ragged rows (see gemm_nt_w384_kernel): fragment jobs after the block's last tile (not in the derivative instantiation:
  // with its aux_in epilogue the section costs the 256-register kernel 12 bytes of scratch; dfc2 keeps its separate tail launch) ----
  if constexpr (ACT != LC2IS_ACT_DQUICK_GELU && ACT != LC2IS_ACT_DRELU)
  if (p.tail_rows > 0) {
    GemmNtArgs q = p;
    q.M = p.M + p.tail_rows;
    const int nfn = p.N / 16, njobs = nfn * ((p.tail_rows + 15) / 16);
    for (int job = wid * (int)gridDim.x + (int)blockIdx.x; job < njobs; job += 8 * (int)gridDim.x)
      rows_fragment<16, ACT>(q, (job % nfn) * 16, p.M + (job / nfn) * 16, lane0);
  }
}

template <int ACT>
int launch_pp_act(const GemmNtArgs& a, hipStream_t stream) {
  constexpr int LDS = (256 + 256) * 128 + 8 * 64 * 144;
  static DevOnce attr_set;
  if (attr_set.need()) {
    if (hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return LC2IS_ERR_LAUNCH;
    attr_set.done();
  }
  const int ntiles = ((a.M + 255) / 256) * ((a.N + 255) / 256);
  const int grid = ntiles < lc2is_ncu() ? ntiles : lc2is_ncu();   // one block per CU of the budget (common.h)
  hipLaunchKernelGGL(gemm_nt_pp_kernel<ACT>, dim3(grid), dim3(512), LDS, stream, a, ntiles);
  return lc2is_check_launch();
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch_cfg(const GemmNtArgs& a, hipStream_t stream, int batch = 1) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int LDS = 2 * (BM + BN) * 128;
  auto kern = gemm_nt_kernel<BM, BN, WAVES_M, WAVES_N>;
  static DevOnce attr_set;  // idempotent; a race only repeats the same call
  if (attr_set.need()) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return LC2IS_ERR_LAUNCH;
    attr_set.done();
  }
  const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL(kern, dim3(ntm * ntn, batch), dim3(NT), LDS, stream, a);
  return lc2is_check_launch();
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI = -1>
int launch_dma(const GemmNtArgs& a, hipStream_t stream) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int STAGES = 2 * (BM + BN) * 128, PATCHES = EPI == -2 ? WAVES_M * WAVES_N * 64 * 272 : 0;   // fp32 epilogue patches overlay the stages
  constexpr int LDS = STAGES > PATCHES ? STAGES : PATCHES;
  auto kern = gemm_nt_dma_kernel<BM, BN, WAVES_M, WAVES_N, EPI>;
  static DevOnce attr_set;
  if (attr_set.need()) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return LC2IS_ERR_LAUNCH;
    attr_set.done();
  }
  const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL(kern, dim3(ntm * ntn), dim3(NT), LDS, stream, a);
  return lc2is_check_launch();
}

// what the 256x384 kernel takes: N a multiple of 384, no activation, and either fp32-only output (bias, fp32 residual) or
// bf16-only output (bias); 32-bit byte offsets into the outputs
bool w384_f32_ok(const GemmNtArgs& a) {
  return a.N % 384 == 0 && a.act == LC2IS_ACT_NONE && a.out_f32 && !a.out_bf16 && !a.aux_out && !a.aux_in &&
         (double)a.M * a.ldf * 4.0 < 2147483648.0 && (!a.resid || (double)a.M * a.ldr * 4.0 < 2147483648.0);
}
bool w384_bf16_ok(const GemmNtArgs& a) {
  const bool plain = a.act == LC2IS_ACT_NONE && !a.aux_in;
  const bool resid = a.act == LC2IS_ACT_ADD_AUX && a.aux_in && a.ldx % 8 == 0 && (double)a.M * a.ldx * 2.0 < 2147483648.0;   // bf16 residual stream
  return a.N % 384 == 0 && (plain || resid) && a.out_bf16 && !a.out_f32 && !a.resid && !a.aux_out &&
         a.ldo % 8 == 0 && (double)a.M * a.ldo * 2.0 < 2147483648.0;
}

template <int EPI>
int launch_w384_epi(const GemmNtArgs& a, hipStream_t stream) {
  constexpr int LDS = 2 * (256 + 384) * 128;   // 160 KiB: the whole CU
  auto kern = gemm_nt_w384_kernel<EPI>;
  static DevOnce attr_set;
  if (attr_set.need()) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return LC2IS_ERR_LAUNCH;
    attr_set.done();
  }
  hipLaunchKernelGGL(kern, dim3(((a.M + 255) / 256) * (a.N / 384)), dim3(512), LDS, stream, a);
  return lc2is_check_launch();
}

int launch_w384(const GemmNtArgs& a, hipStream_t stream) {
  if (w384_f32_ok(a)) return launch_w384_epi<-2>(a, stream);
  if (w384_bf16_ok(a)) return a.act == LC2IS_ACT_ADD_AUX ? launch_w384_epi<1>(a, stream) : launch_w384_epi<0>(a, stream);
  return LC2IS_ERR_UNSUPPORTED;
}

int launch_rows(const GemmNtArgs& a, hipStream_t stream) {
  if (a.M > 64 || a.N % 4) return LC2IS_ERR_UNSUPPORTED;
  if (a.K >= 2048) hipLaunchKernelGGL(gemm_nt_rows_kernel<28>, dim3((a.N + 15) / 16, (a.M + 15) / 16), dim3(64), 0, stream, a);   // long K: 56 loads in flight per wave (the 6-bit vmcnt allows 63)
  else hipLaunchKernelGGL(gemm_nt_rows_kernel<16>, dim3((a.N + 15) / 16, (a.M + 15) / 16), dim3(64), 0, stream, a);
  return lc2is_check_launch();
}

// the counted-wait persistent kernel takes bf16-output problems through the staged epilogue only (see the kernel's comment)
bool persist2_ok(const GemmNtArgs& a) {
  const double lim = 2147483648.0;
  return a.staged_epi == 1 && a.out_bf16 && !a.out_f32 && !a.resid && a.K % 64 == 0 && (double)a.M * a.ldo * 2.0 < lim &&
         (!a.aux_out || (double)a.M * a.ldy * 2.0 < lim) && (!a.aux_in || (double)a.M * a.ldx * 2.0 < lim);
}

int launch_pp(const GemmNtArgs& a, hipStream_t stream) {
  if (!persist2_ok(a)) return LC2IS_ERR_UNSUPPORTED;
  switch (a.act) {
    case LC2IS_ACT_QUICK_GELU: return launch_pp_act<LC2IS_ACT_QUICK_GELU>(a, stream);
    case LC2IS_ACT_DQUICK_GELU: return launch_pp_act<LC2IS_ACT_DQUICK_GELU>(a, stream);
    case LC2IS_ACT_NONE: return launch_pp_act<LC2IS_ACT_NONE>(a, stream);
    case LC2IS_ACT_ADD_AUX: return launch_pp_act<LC2IS_ACT_ADD_AUX>(a, stream);
    case LC2IS_ACT_RELU: return launch_pp_act<LC2IS_ACT_RELU>(a, stream);     // (round 5: the decoders' linear1 / its dgrad at >= 2 rounds of tiles)
    case LC2IS_ACT_DRELU: return launch_pp_act<LC2IS_ACT_DRELU>(a, stream);
    default: return LC2IS_ERR_UNSUPPORTED;
  }
}

int launch_by_cfg(const GemmNtArgs& a_in, int cfg, hipStream_t stream) {
  GemmNtArgs a = a_in;
  const bool f32_staged = a.staged_epi == 2;
  if (f32_staged) a.staged_epi = 0;
  switch (cfg) {
    case 1: return launch_cfg<128, 128, 2, 2>(a, stream);
    case 2: return launch_cfg<256, 128, 4, 2>(a, stream);
    case 3: return launch_cfg<64, 64, 2, 2>(a, stream);
    case 4:
      if (f32_staged) return launch_dma<256, 256, 2, 4, -2>(a, stream);
      switch (a.act) {
        case LC2IS_ACT_QUICK_GELU: return launch_dma<256, 256, 2, 4, LC2IS_ACT_QUICK_GELU>(a, stream);
        case LC2IS_ACT_RELU: return launch_dma<256, 256, 2, 4, LC2IS_ACT_RELU>(a, stream);
        case LC2IS_ACT_DQUICK_GELU: return launch_dma<256, 256, 2, 4, LC2IS_ACT_DQUICK_GELU>(a, stream);
        case LC2IS_ACT_DRELU: return launch_dma<256, 256, 2, 4, LC2IS_ACT_DRELU>(a, stream);
        case LC2IS_ACT_GELU_ERF: return launch_dma<256, 256, 2, 4, LC2IS_ACT_GELU_ERF>(a, stream);
        case LC2IS_ACT_DGELU_ERF: return launch_dma<256, 256, 2, 4, LC2IS_ACT_DGELU_ERF>(a, stream);
        case LC2IS_ACT_NONE: return launch_dma<256, 256, 2, 4, LC2IS_ACT_NONE>(a, stream);
        case LC2IS_ACT_ADD_AUX: return launch_dma<256, 256, 2, 4, LC2IS_ACT_ADD_AUX>(a, stream);
        default: return launch_dma<128, 128, 2, 2>(a, stream);   // codes 5 / 6 (the save-the-derivative experiment): the 128x128 kernel's run-time switch
                                                                 // (the 256x256 instantiation of it spilled 128 bytes and left the library in round 5)
      }
    case 6:
      if (f32_staged) return launch_dma<128, 128, 2, 2, -2>(a, stream);
      switch (a.act) {   // the activations of the Swin / hierarchical-decoder GEMMs that land on this tile size
        case LC2IS_ACT_RELU: return launch_dma<128, 128, 2, 2, LC2IS_ACT_RELU>(a, stream);
        case LC2IS_ACT_DRELU: return launch_dma<128, 128, 2, 2, LC2IS_ACT_DRELU>(a, stream);
        case LC2IS_ACT_GELU_ERF: return launch_dma<128, 128, 2, 2, LC2IS_ACT_GELU_ERF>(a, stream);
        case LC2IS_ACT_DGELU_ERF: return launch_dma<128, 128, 2, 2, LC2IS_ACT_DGELU_ERF>(a, stream);
        case LC2IS_ACT_NONE: return launch_dma<128, 128, 2, 2, LC2IS_ACT_NONE>(a, stream);
        case LC2IS_ACT_ADD_AUX: return launch_dma<128, 128, 2, 2, LC2IS_ACT_ADD_AUX>(a, stream);
        default: return launch_dma<128, 128, 2, 2>(a, stream);
      }
    case 15: return launch_pp(a, stream);
    case 16: return launch_w384(a, stream);
    case 17: return launch_rows(a, stream);
    default: return LC2IS_ERR_UNSUPPORTED;
  }
}

}  // namespace

extern "C" int lc2is_gemm_nt_bf16(const void* A, int lda, const void* W, int ldw, const float* bias,
                                  const float* resid, int ldr, const void* aux_in, int ldx, void* out_bf16,
                                  int ldo, float* out_f32, int ldf, void* aux_out, int ldy, int M, int N,
                                  int K, int act, int tile_cfg, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!A || !W) return LC2IS_ERR_NULL;
  if (!out_bf16 && !out_f32) return LC2IS_ERR_NULL;
  if (M <= 0 || N <= 0 || K <= 0) return LC2IS_ERR_SHAPE;
  if (K % 64 != 0 || N % 4 != 0) return LC2IS_ERR_SHAPE;
  if (lda < K || ldw < K || lda % 8 || ldw % 8) return LC2IS_ERR_SHAPE;
  if ((out_bf16 && (ldo < N || ldo % 4)) || (out_f32 && (ldf < N || ldf % 4)) || (resid && (ldr < N || ldr % 4)) ||
      (aux_in && (ldx < N || ldx % 4)) || (aux_out && (ldy < N || ldy % 4)))
    return LC2IS_ERR_SHAPE;
  if ((act == LC2IS_ACT_DQUICK_GELU || act == LC2IS_ACT_DRELU || act == LC2IS_ACT_MUL_AUX || act == LC2IS_ACT_DGELU_ERF ||
       act == LC2IS_ACT_ADD_AUX) && !aux_in)
    return LC2IS_ERR_NULL;
  if (act < LC2IS_ACT_NONE || act > LC2IS_ACT_ADD_AUX) return LC2IS_ERR_UNSUPPORTED;
  // 32-bit buffer offsets: operand panels (plus one tile of overhang) must stay under 2 GiB
  if ((double)(M + 256) * lda * 2.0 >= 2147483648.0 || (double)(N + 256) * ldw * 2.0 >= 2147483648.0)
    return LC2IS_ERR_UNSUPPORTED;
  GemmNtArgs a{(const bf16_t*)A, lda, (const bf16_t*)W, ldw, bias, resid, ldr, (const bf16_t*)aux_in, ldx,
               (bf16_t*)out_bf16, ldo, out_f32, ldf, (bf16_t*)aux_out, ldy, M, N, K, act, 0};
  a.staged_epi = (N % 8 == 0) && (!out_bf16 || ldo % 8 == 0) && (!aux_out || ldy % 8 == 0) &&
                 (!aux_in || ldx % 8 == 0) && (out_bf16 || aux_out || aux_in);
  static const bool f32_lds = !(getenv("LC2IS_F32_STAGED") && atoi(getenv("LC2IS_F32_STAGED")) == 0);
  if (f32_lds && out_f32 && !out_bf16 && !aux_out && !aux_in && act == LC2IS_ACT_NONE &&
      (double)M * ldf * 4.0 < 2147483648.0 && (!resid || (double)M * ldr * 4.0 < 2147483648.0))
    a.staged_epi = 2;
  int cfg = tile_cfg;
  if (cfg != 0) return launch_by_cfg(a, cfg, stream);
  // ragged-row tails (<= 64 rows) of the exact-round plans: the one-wave-per-fragment kernel (cfg 17); LC2IS_GEMM_ROWS_TAIL=0: 64x64 tiles
  static const int tail_cfg = (getenv("LC2IS_GEMM_ROWS_TAIL") && atoi(getenv("LC2IS_GEMM_ROWS_TAIL")) == 0) ? 3 : 17;
  // ... and behind a 256x384 launch they are folded into it (fragment jobs after each block's own tile); LC2IS_GEMM_TAIL_FOLD=0: separate launch
  static const bool fold_tail = tail_cfg == 17 && !(getenv("LC2IS_GEMM_TAIL_FOLD") && atoi(getenv("LC2IS_GEMM_TAIL_FOLD")) == 0);
  const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128);
  static const long cfg6_min = getenv("LC2IS_GEMM_CFG6_MIN") ? atol(getenv("LC2IS_GEMM_CFG6_MIN")) : 128;   // (512 and the register-staged 128x128 kernel below it measured 0.8 % slower on config 5)
  // The large-tile (256-row) family takes a problem from about one round of 256x256 tiles on: >= 1024 tiles of 128x128, or (round 5)
  // >= 7/8 of a round of 256x256 tiles — ViT-L/14 at B = 8 (M = 16 208, N = 1024: 127 x 8 = 1016 small tiles, 64 x 4 = 256 large ones,
  // exactly ONE round) sat a hair under the first rule and ran its N = 1024 GEMMs on 128x128 tiles at 379 TF/s (profiles/r05_config4_*).
  static const long big_min256 = getenv("LC2IS_GEMM_BIG_MIN256") ? atol(getenv("LC2IS_GEMM_BIG_MIN256")) : 224;
  const long tiles256 = (long)((M + 255) / 256) * (N / 256);
  if (!((tiles128 >= 1024 || tiles256 >= big_min256) && N % 256 == 0)) {
    if (tiles128 >= cfg6_min) cfg = 6;              // 128x128 LDS-DMA tiles, 2 blocks/CU
    else if (tiles128 >= 128) cfg = 1;
    else cfg = 3;                                        // small problem: 64x64 tiles to fill the chip
    return launch_by_cfg(a, cfg, stream);
  }
  // cfg 16 (256x384 tiles): N = 768 at M = 128 x 256 is exactly one round instead of 1.5 (LC2IS_GEMM_W384=0: off)
  static const bool use_w384 = !(getenv("LC2IS_GEMM_W384") && atoi(getenv("LC2IS_GEMM_W384")) == 0);
  static const bool use_w384_bf16 = !(getenv("LC2IS_GEMM_W384_BF16") && atoi(getenv("LC2IS_GEMM_W384_BF16")) == 0);
  if (use_w384 && use_w384_bf16 && w384_bf16_ok(a)) {   // bf16-only, no activation (dqkv, dfc1, dout_proj): take it where it saves >= 8 % of the tile time
    const int r = M % 256;
    const bool can_peel = r > 0 && r <= 64 && M > 256;
    const int mm = can_peel ? M - r : M;
    const long t384 = (long)((mm + 255) / 256) * (N / 384), t256 = (long)((mm + 255) / 256) * (N / 256);
    if ((double)lc2is_rounds(t384) * 1.5 < 0.92 * (double)lc2is_rounds(t256)) {   // (qkv, N = 2304: 3 rounds of 384 = 4.5 against 5 tile times: 117 vs 123 us)
      GemmNtArgs main_part = a, tail = a;
      main_part.M = mm;
      if (fold_tail) main_part.tail_rows = M - mm;   // the ragged rows ride in the same launch (rows M .. follow in the same buffers)
      int rc = launch_by_cfg(main_part, 16, stream);
      if (rc || mm == M || fold_tail) return rc;
      tail.M = M - mm;
      tail.A = a.A + (size_t)mm * lda;
      tail.out_bf16 = a.out_bf16 + (size_t)mm * ldo;
      if (a.aux_in) tail.aux_in = a.aux_in + (size_t)mm * ldx;
      return launch_by_cfg(tail, tail_cfg, stream);
    }
  }
  // bf16-output problems of >= 2 rounds of tiles: the persistent form with counted waits across the tile seam (cfg 13; ragged
  // rows stay in the launch — its blocks walk the tiles, there is no round to save): -4..10 % on the K = 768 shapes
  static const bool use_persist = !(getenv("LC2IS_GEMM_PERSIST") && atoi(getenv("LC2IS_GEMM_PERSIST")) == 0);
  static const long persist_min = getenv("LC2IS_GEMM_PERSIST_MIN") ? atol(getenv("LC2IS_GEMM_PERSIST_MIN")) : 257;   // more than one round of tiles (A/B 512 -> 257: 907 -> 915 img/s)
  if (use_persist && persist2_ok(a) && (long)((M + 255) / 256) * (N / 256) >= persist_min &&
      (act == LC2IS_ACT_NONE || act == LC2IS_ACT_QUICK_GELU || act == LC2IS_ACT_DQUICK_GELU || act == LC2IS_ACT_ADD_AUX ||
       act == LC2IS_ACT_RELU || act == LC2IS_ACT_DRELU)) {
    // One block per CU walks tiles t, t + 256, ...: the launch lasts ceil(tiles / 256) tile times.  The ragged last <= 64 rows
    // (B x 1025 tokens: 32 rows, i.e. one more row of N / 256 tiles) are peeled off into a small-tile launch when that saves a
    // whole tile time: fc1 / dfc2 at M = 32 800 walk 1548 tiles = 6 rounds + 12 tiles, 1536 = exactly 6 without the 32 rows.
    static const bool peel = !(getenv("LC2IS_GEMM_PERSIST_PEEL") && atoi(getenv("LC2IS_GEMM_PERSIST_PEEL")) == 0);
    constexpr int pcfg = 15;   // the wave groups in ping-pong (round 4: 966 -> 975 img/s against cfg 13, the lockstep persistent form — now tools/probes/gemm_nt_persist2/)
    const int r = M % 256;
    const long tiles_all = (long)((M + 255) / 256) * (N / 256), tiles_main = (long)(M / 256) * (N / 256);
    if (peel && r > 0 && r <= 64 && M > 256 && lc2is_rounds(tiles_main) < lc2is_rounds(tiles_all)) {
      GemmNtArgs main_part = a, tail = a;
      main_part.M = M - r;
      if (fold_tail && pcfg == 15 && N % 16 == 0 && act != LC2IS_ACT_DQUICK_GELU && act != LC2IS_ACT_DRELU) main_part.tail_rows = r;
      int rc = launch_by_cfg(main_part, pcfg, stream);
      if (rc || main_part.tail_rows) return rc;
      const size_t m0 = (size_t)(M - r);
      tail.M = r;
      tail.A = a.A + m0 * lda;
      if (a.aux_in) tail.aux_in = a.aux_in + m0 * ldx;
      if (a.out_bf16) tail.out_bf16 = a.out_bf16 + m0 * ldo;
      if (a.aux_out) tail.aux_out = a.aux_out + m0 * ldy;
      return launch_by_cfg(tail, tail_cfg, stream);
    }
    return launch_by_cfg(a, pcfg, stream);
  }
  // Plenty of work: 256x256 LDS-DMA tiles, one block per CU, so time = rounds x tile cost; the ragged last <= 64 rows
  // (B x 1025 tokens: 32 rows) are peeled off into a small-tile launch when that saves a whole round of tiles.
  int best_cfg = 0, best_main = M;
  double best_cost = 1e300;
  for (int split = 0; split < 2; ++split)
    for (int c : {4, 16}) {   // (128x384 measured ~45 % slower per flop and left the library)
      const int bm = 256, bn = c == 16 ? 384 : 256;
      if (c == 16 && !(use_w384 && a.staged_epi == 2 && w384_f32_ok(a))) continue;   // (fp32-only output here; the bf16 form is chosen above)
      if (N % bn) continue;
      const int r = M % bm;
      if (split && (r == 0 || r > 64 || M <= bm)) continue;
      const int mm = split ? M - r : M;
      const long tiles = (long)((mm + bm - 1) / bm) * (N / bn);
      double cost = (double)lc2is_rounds(tiles) * bm * bn;
      if (split) cost += 65536.0 * 0.3 * 768.0 / K;                                      // ~8 us for the extra launch
      if (split) cost *= 1.03;                                                            // prefer the plain plan on near-ties
      if (cost < best_cost) { best_cost = cost; best_cfg = c; best_main = mm; }
    }
  if (best_main == M) return launch_by_cfg(a, best_cfg, stream);
  GemmNtArgs main_part = a, tail = a;
  main_part.M = best_main;
  if (best_cfg == 16 && fold_tail) main_part.tail_rows = M - best_main;
  int rc = launch_by_cfg(main_part, best_cfg, stream);
  if (rc || main_part.tail_rows) return rc;
  const size_t m0 = (size_t)best_main;
  tail.M = M - best_main;
  tail.A = a.A + m0 * lda;
  if (a.resid) tail.resid = a.resid + m0 * ldr;
  if (a.aux_in) tail.aux_in = a.aux_in + m0 * ldx;
  if (a.out_bf16) tail.out_bf16 = a.out_bf16 + m0 * ldo;
  if (a.out_f32) tail.out_f32 = a.out_f32 + m0 * ldf;
  if (a.aux_out) tail.aux_out = a.aux_out + m0 * ldy;
  return launch_by_cfg(tail, tail_cfg, stream);
}

// out_f32 = A.W^T + bias (+ resid) AND ln_out = bf16(LayerNorm(out_f32) * gamma + beta) (+ mean / rstd) in ONE launch of the
// 256x384-tile kernel (EPI = -3: the two column tiles of a row tile exchange their row statistics through `xchg`); N = 384 or 768.
// The <= 64 ragged rows of B x 1025-token inputs ride in the launch as fragment jobs (fp32 output), their LayerNorm is a second,
// small launch of the LayerNorm kernel.  LC2IS_ERR_UNSUPPORTED: the caller runs lc2is_gemm_nt_bf16 + lc2is_layernorm_fwd instead.
extern "C" size_t lc2is_gemm_nt_ln_xchg_bytes(int M, int N) { return N == 768 && M > 0 ? (size_t)M * 32 : 0; }

extern "C" int lc2is_gemm_nt_ln_bf16(const void* A, int lda, const void* W, int ldw, const float* bias, const float* resid,
                                     int ldr, float* out_f32, int ldf, const float* gamma, const float* beta, float eps,
                                     void* ln_out, int ldl, float* mean, float* rstd, void* xchg, size_t xchg_bytes, int M,
                                     int N, int K, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!A || !W || !out_f32 || !ln_out || !gamma) return LC2IS_ERR_NULL;
  if (M <= 0 || N <= 0 || K <= 0) return LC2IS_ERR_SHAPE;
  if (K % 64 != 0 || lda < K || ldw < K || lda % 8 || ldw % 8) return LC2IS_ERR_SHAPE;
  if (ldf < N || ldf % 4 || (resid && (ldr < N || ldr % 4)) || ldl < N || ldl % 8) return LC2IS_ERR_SHAPE;
  if (N != 384 && N != 768) return LC2IS_ERR_UNSUPPORTED;
  if (M < 256) return LC2IS_ERR_UNSUPPORTED;
  const double lim = 2147483648.0;
  if ((double)(M + 256) * lda * 2.0 >= lim || (double)(N + 256) * ldw * 2.0 >= lim || (double)M * ldf * 4.0 >= lim ||
      (resid && (double)M * ldr * 4.0 >= lim) || (double)M * ldl * 2.0 >= lim)
    return LC2IS_ERR_UNSUPPORTED;
  if (N == 768 && (!xchg || xchg_bytes < lc2is_gemm_nt_ln_xchg_bytes(M, N) || ((uintptr_t)xchg & 7))) return LC2IS_ERR_WORKSPACE;
  const int r = M % 256;
  if (r > 64) return LC2IS_ERR_UNSUPPORTED;   // (the tiles of the fused form are whole: the ragged rows are at most the 64 the fragment jobs take)
  const bool peel = r > 0;
  const int mm = M - r;
  GemmNtArgs a{(const bf16_t*)A, lda, (const bf16_t*)W, ldw, bias, resid, ldr, nullptr, 0, nullptr, 0, out_f32, ldf, nullptr, 0,
               mm, N, K, LC2IS_ACT_NONE, 0};
  a.tail_rows = peel ? r : 0;
  a.ln_gamma = gamma; a.ln_beta = beta;
  a.ln_out = (bf16_t*)ln_out; a.ldl = ldl;
  a.ln_mean = mean; a.ln_rstd = rstd;
  a.ln_xchg = (unsigned long long*)xchg;
  a.ln_eps = eps;
  int rc = launch_w384_epi<-3>(a, stream);
  if (rc || !peel) return rc;
  return lc2is_layernorm_fwd(out_f32 + (size_t)mm * ldf, ldf, 0, gamma, beta, (bf16_t*)ln_out + (size_t)mm * ldl, ldl, nullptr, 0,
                             mean ? mean + mm : nullptr, rstd ? rstd + mm : nullptr, r, N, eps, stream_);
}

// Strided-batched plain product: out[b] = A[b] (M x K) . W[b]^T (N x K), b = 0..batch-1, in ONE launch (blockIdx.y = b).
// replaces: torch.einsum('bchw,bkc->bkhw', visual, text) — one class-embedding matrix PER IMAGE (reference
// model/final.py:355, model/model.py:161,210, model/ftn.py:60) — and its two backward contractions.
extern "C" int lc2is_gemm_nt_bf16_batched(const void* A, int lda, long stride_a, const void* W, int ldw, long stride_w,
                                          void* out_bf16, int ldo, long stride_ob, float* out_f32, int ldf,
                                          long stride_of, int M, int N, int K, int batch, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!A || !W) return LC2IS_ERR_NULL;
  if (!out_bf16 && !out_f32) return LC2IS_ERR_NULL;
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || batch > 65535) return LC2IS_ERR_SHAPE;
  if (K % 64 != 0 || N % 4 != 0) return LC2IS_ERR_SHAPE;
  if (lda < K || ldw < K || lda % 8 || ldw % 8 || stride_a % 8 || stride_w % 8) return LC2IS_ERR_SHAPE;
  if ((out_bf16 && (ldo < N || ldo % 4 || stride_ob % 4)) || (out_f32 && (ldf < N || ldf % 4 || stride_of % 4)))
    return LC2IS_ERR_SHAPE;
  if ((double)(M + 256) * lda * 2.0 >= 2147483648.0 || (double)(N + 256) * ldw * 2.0 >= 2147483648.0)
    return LC2IS_ERR_UNSUPPORTED;
  GemmNtArgs a{(const bf16_t*)A, lda, (const bf16_t*)W, ldw, nullptr, nullptr, 0, nullptr, 0,
               (bf16_t*)out_bf16, ldo, out_f32, ldf, nullptr, 0, M, N, K, LC2IS_ACT_NONE, 0,
               stride_a, stride_w, stride_ob, stride_of};
  const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128) * batch;
  if (tiles128 >= 128) return launch_cfg<128, 128, 2, 2>(a, stream, batch);
  return launch_cfg<64, 64, 2, 2>(a, stream, batch);
}
