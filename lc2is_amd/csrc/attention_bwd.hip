// Fused attention backward (flash-style: P is recomputed from Q, K and the forward's log-sum-exp).
// replaces: autograd of hf eager_attention_forward (hf:modeling_clip.py:259-277) and of torch
//   multi_head_attention_forward's attention core (model/decoder.py:9-21), reached from loss.backward()
//   (reference engine.py:100).
//
// Two launches, no atomics, bitwise reproducible:
//   (1) dQ kernel — "query on the lane" (same orientation as the forward): per 64-key tile
//       S^T = K·Q^T, dP^T = V·dO^T, dS^T = P^T∘(dP^T − δ), dQ^T[d][q] += K^T·dS^T.  δ = rowsum(dO∘O) is
//       computed in this kernel's prologue (each lane owns one query row) and written out for (2).
//   (2) dK/dV kernel — "key on the lane": a wave owns 32 keys (K, V fragments live in registers for the whole
//       kernel), and sweeps the queries: S = Q·K^T, dP = dO·V^T, then dV^T[d][key] += dO^T·P and
//       dK^T[d][key] += Q^T·dS take the fp32 accumulators of S/dP, packed to bf16, directly as MFMA B
//       operands (accumulator-as-operand), so nothing but the shared Q/dO tiles touches LDS (they arrive by LDS-DMA,
//       double-buffered, like the K / V tiles of (1)).
// The shared tiles are read BOTH by rows (ds_read_b128, MFMA A operand of S/dP) and by columns
// (ds_read_b64_tr_b16, A operand of the transposed products) from ONE LDS image, made conflict-free for
// both by an XOR swizzle of the 16-byte chunk index (128-byte rows for D=64, 256-byte rows otherwise).
#include "attn_common.h"
#include "lc2is_hip.h"
#include <cstdlib>

namespace {

struct AttnBwdArgs {
  const bf16_t* Q; int ldq;
  const bf16_t* K; int ldk;
  const bf16_t* V; int ldv;
  const bf16_t* O; int ldo;
  const bf16_t* dO; int lddo;
  bf16_t* dQ; int lddq;
  bf16_t* dK; int lddk;
  bf16_t* dV; int lddv;
  const float* lse2;   // [B,H,Sq]
  float* delta;        // [B,H,Sq]  written by the dQ kernel, read by the dK/dV kernel
  const float* kbias;  // [B,Sk] or null
  int B, H, Sq, Sk;
  float scale, scale_log2;
  int causal;
  DropCfg drop;        // attention-probability dropout of the forward (DROP instantiations only)
};

constexpr float LOG2E = 1.44269504088896341f;

template <int D> struct Img {
  static constexpr int PITCH = (D == 64) ? 128 : 256;
  static constexpr int TILE = 64 * PITCH;
  static constexpr int CH = D / 8;
  static constexpr int NCH = 64 * CH / 256;
  // byte offset of 16-byte chunk `ch` of row `row`
  __device__ static __forceinline__ int off(int row, int ch) {
    if constexpr (D == 64) {
      return row * 128 + ((ch ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3))) << 4);
    } else {
      return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
    }
  }
};

__device__ __forceinline__ bf16x8_t tr_frag3(const char* base, int addr_lo, int addr_hi) {
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4_t*)LDS_PTR(base + addr_lo));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4_t*)LDS_PTR(base + addr_hi));
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

__device__ __forceinline__ bf16x8_t pack8(const f32x16_t& v, int base) {
  bf16x8_t r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)v[base + j];
  return r;
}

__device__ __forceinline__ float dot8(const i32x4_t& a, const i32x4_t& b) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned x = (unsigned)a[i], y = (unsigned)b[i];
    s += bf16_to_f32((bf16_t)(x & 0xffff)) * bf16_to_f32((bf16_t)(y & 0xffff));
    s += bf16_to_f32((bf16_t)(x >> 16)) * bf16_to_f32((bf16_t)(y >> 16));
  }
  return s;
}

// ------------------------------------------------------------------------------------------------------
// (1) dQ kernel: grid (ceil(Sq/128), H, B), 4 waves x 32 queries; loops over 64-key tiles (the forward kernel's structure):
// K / V tiles arrive by LDS-DMA into a ring behind counted s_waitcnt vmcnt + one raw s_barrier per tile (no staging registers,
// no ds_write pass), and the S^T / dP^T chains of the next 32-key half tile are issued in the same basic block as the exp2 / dS
// arithmetic and the dQ products of the current one.  Interior tiles run unrolled over the ring (stage offsets are instruction
// immediates) without mask code; tail / causal-edge / key-bias tiles take the rolled general body.  dQ rows leave through LDS
// as whole lines.
// ------------------------------------------------------------------------------------------------------
// DROP (both kernels): the forward dropped attention probabilities, O = (keep∘P / (1-p)) V.  Then dP = keep∘(dO V^T) / (1-p),
// dV = (keep∘P / (1-p))^T dO, dS = P∘(dP − δ) with the UNDROPPED P and δ = rowsum(dO∘O) as before; keep is re-evaluated
// from the same (row, key) coordinates as in attention_fwd.hip.
constexpr int LC2IS_DQ2_WAVES = 3;      // waves per SIMD the dQ kernel is compiled for at D = 64 (2 measured no slower)
constexpr int LC2IS_DQ2_DP_AHEAD = 1;   // the dP^T chain of the next half tile is issued ahead together with its S^T chain (16 more live registers)
template <int D, bool DROP = false>
__global__ __launch_bounds__(256, (D == 64) ? LC2IS_DQ2_WAVES : (D <= 96 ? 2 : 1)) void attn_bwd_dq2_kernel(AttnBwdArgs p) {
  using Cfg = AttnCfg<D>;
  constexpr int PITCH = Cfg::PITCH, NSTAGE = Cfg::NSTAGE, PD = NSTAGE - 1, STG = Cfg::STAGE;
  constexpr int NKS = D / 16, NDT = D / 32;
  constexpr int OP = 2 * D + 16;        // row pitch of the output staging image (bytes)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, l31 = lane & 31;
  const int nqb = (p.Sq + 127) / 128;   // 1-D XCD-aware grid (see attention_fwd.hip)
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int bx = tile % nqb, head = (tile / nqb) % p.H, b = tile / (nqb * p.H);
  const int q0 = bx * 128 + wid * 32;
  const int qrow = q0 + l31;
  const bool qok = qrow < p.Sq;
  const bool wave_active = q0 < p.Sq;   // wave-uniform
  const float INF = __builtin_inff();
  const unsigned drop_rh = DROP ? drop_row_hash(p.drop, (unsigned)((b * p.H + head) * p.Sq + qrow)) : 0u;

  int nkt = (p.Sk + 63) / 64;
  if (p.causal) {
    const int lim = (bx * 128 + 128 + 63) / 64;
    if (lim < nkt) nkt = lim;
  }

  const unsigned kbytes = (unsigned)p.B * p.Sk * p.ldk * 2u, vbytes = (unsigned)p.B * p.Sk * p.ldv * 2u;
  int k_goff[Cfg::PPW], v_goff[Cfg::PPW];
#pragma unroll
  for (int j = 0; j < Cfg::PPW; ++j) {
    const int row = Cfg::RPP * (wid * Cfg::PPW + j) + lane / Cfg::SLOTS;
    const int ch = (lane % Cfg::SLOTS) ^ Cfg::swz(row);
    k_goff[j] = ch < Cfg::CH ? ((b * p.Sk + row) * p.ldk + head * D + ch * 8) * 2 : -1;
    v_goff[j] = ch < Cfg::CH ? ((b * p.Sk + row) * p.ldv + head * D + ch * 8) * 2 : -1;
  }
  auto request = [&](int kt, int so) __attribute__((always_inline)) {
    attn_dma_tile<D>(p.K, kbytes, p.V, vbytes, smem + so, wid, k_goff, v_goff, kt * 64 * p.ldk * 2, kt * 64 * p.ldv * 2);
  };
#pragma unroll
  for (int i = 0; i < PD; ++i)
    if (i < nkt) request(i, i * STG);

  // Q / dO fragments (B operands), delta = rowsum(dO . O) (written for the dK/dV kernel), the row's log-sum-exp
  bf16x8_t qf[NKS], dof[NKS];
  float dpart = 0.f;
  {
    const __amdgpu_buffer_rsrc_t rsQ = make_rsrc(p.Q, (unsigned)p.B * p.Sq * p.ldq * 2u);
    const __amdgpu_buffer_rsrc_t rsO = make_rsrc(p.O, (unsigned)p.B * p.Sq * p.ldo * 2u);
    const __amdgpu_buffer_rsrc_t rsdO = make_rsrc(p.dO, (unsigned)p.B * p.Sq * p.lddo * 2u);
    const int tok = b * p.Sq + qrow;
    const int qo = (tok * p.ldq + head * D + 8 * hh) * 2;
    const int oo = (tok * p.ldo + head * D + 8 * hh) * 2;
    const int go = (tok * p.lddo + head * D + 8 * hh) * 2;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      const i32x4_t qv = __builtin_amdgcn_raw_buffer_load_b128(rsQ, qok ? qo + s * 32 : -1, 0, 0);
      const i32x4_t ov = __builtin_amdgcn_raw_buffer_load_b128(rsO, qok ? oo + s * 32 : -1, 0, 0);
      const i32x4_t gv = __builtin_amdgcn_raw_buffer_load_b128(rsdO, qok ? go + s * 32 : -1, 0, 0);
      qf[s] = __builtin_bit_cast(bf16x8_t, qv);
      dof[s] = __builtin_bit_cast(bf16x8_t, gv);
      dpart += dot8(ov, gv);
    }
  }
  const float delta = dpart + __shfl_xor(dpart, 32, 64);
  float lse = INF;  // padded query rows: exp2(x - inf) = 0
  if (qok) {
    const size_t si = ((size_t)b * p.H + head) * p.Sq + qrow;
    const float l = p.lse2[si];
    lse = (l == -INF) ? INF : l;
    if (hh == 0) p.delta[si] = delta;
  }

  // opaque LDS fragment bases in stage 0 (attention_fwd.hip): K rows (V rows = + TILE), K transposed
  const int q4 = (lane >> 2) & 3, p4 = lane & 3, cg = (lane >> 4) & 1;
  const unsigned smem_a = (unsigned)(size_t)LDS_PTR(smem);
  unsigned k_row[NKS], t_lo[NDT], t_hi[NDT];
#pragma unroll
  for (int s = 0; s < NKS; ++s) {
    k_row[s] = smem_a + Cfg::off(l31, 2 * s + hh);
    asm volatile("" : "+v"(k_row[s]));
  }
#pragma unroll
  for (int d = 0; d < NDT; ++d) {
    const int e = 32 * d + 16 * cg + 4 * p4;
    t_lo[d] = smem_a + Cfg::off(4 * hh + q4, e >> 3) + (e & 7) * 2;
    t_hi[d] = smem_a + Cfg::off(4 * hh + q4 + 8, e >> 3) + (e & 7) * 2;
    asm volatile("" : "+v"(t_lo[d]), "+v"(t_hi[d]));
  }

  f32x16_t dq[NDT];
#pragma unroll
  for (int d = 0; d < NDT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;

  // S^T = K.Q^T and dP^T = V.dO^T of one 32-key half tile
  auto issue_s = [&](int so, int t, f32x16_t& st) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int s = 0; s < NKS; ++s)
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_read_b128(k_row[s] + (unsigned)(so + 32 * t * PITCH)), qf[s], st, 0, 0, 0);
  };
  auto issue_dp = [&](int so, int t, f32x16_t& dpt) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) dpt[r] = 0.f;
#pragma unroll
    for (int s = 0; s < NKS; ++s)
      dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_read_b128(k_row[s] + (unsigned)(so + Cfg::TILE + 32 * t * PITCH)), dof[s], dpt,
                                                    0, 0, 0);
  };
  auto issue = [&](int so, int t, f32x16_t& st, f32x16_t& dpt) __attribute__((always_inline)) {
    issue_s(so, t, st);
    if (LC2IS_DQ2_DP_AHEAD) issue_dp(so, t, dpt);
  };
  // P = exp2(s - lse), dS = P (dP - delta), dQ^T += K^T . dS^T
  auto finish = [&](f32x16_t& st, f32x16_t& dpt, int so, int kt, int t, auto masked_c) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(masked_c)::value;
    if (!LC2IS_DQ2_DP_AHEAD) issue_dp(so, t, dpt);
    if constexpr (DROP) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned key = (unsigned)(kt * 64 + 32 * t + 8 * (r >> 2) + 4 * hh + (r & 3));
        dpt[r] = drop_keep(p.drop, drop_rh, key) ? dpt[r] * p.drop.inv_keep : 0.f;
      }
    }
    const bool tail = (kt * 64 + 64 > p.Sk);
    const bool diag = p.causal && (kt * 64 + 63 > bx * 128);
    if (MASKED && (tail || diag || p.kbias != nullptr)) {   // masks go onto the RAW scores (scale_log2 > 0), then the common arithmetic
      int key0 = kt * 64 + 32 * t + 4 * hh, qr = qrow;
      asm volatile("" : "+v"(key0), "+v"(qr));
      const float inv_sl2 = LOG2E / p.scale_log2;   // key bias in raw-score units
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int key = key0 + 8 * c + j;
          float sc = st[4 * c + j];
          if (key >= p.Sk) sc = -INF;
          else if (p.kbias) sc += p.kbias[(size_t)b * p.Sk + key] * inv_sl2;
          if (diag && key > qr) sc = -INF;
          st[4 * c + j] = sc;
        }
    }
    {
      const f32x2_t sc2 = {p.scale_log2, p.scale_log2}, nl2 = {-lse, -lse}, dl2 = {delta, delta};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2_t s2 = f32x2_t{st[r], st[r + 1]} * sc2 + nl2;
        const f32x2_t pr = {__builtin_amdgcn_exp2f(s2[0]), __builtin_amdgcn_exp2f(s2[1])};
        const f32x2_t ds = pr * (f32x2_t{dpt[r], dpt[r + 1]} - dl2);
        dpt[r] = ds[0];
        dpt[r + 1] = ds[1];
      }
    }
#pragma unroll
    for (int s2i = 0; s2i < 2; ++s2i) {
      const bf16x8_t dsf = pack8(dpt, 8 * s2i);
      const unsigned roff = (unsigned)(so + (32 * t + 16 * s2i) * PITCH);
#pragma unroll
      for (int d = 0; d < NDT; ++d) {
        const bf16x8_t ktf = tr_frag2a(t_lo[d] + roff, t_hi[d] + roff);
        dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, dsf, dq[d], 0, 0, 0);
      }
    }
  };
  auto land = [&]() __attribute__((always_inline)) {
    wait_vm0();
    __builtin_amdgcn_s_barrier();
  };

  int n_plain = p.kbias ? 0 : p.Sk / 64;
  if (p.causal && 2 * bx < n_plain) n_plain = 2 * bx;
  if (nkt < n_plain) n_plain = nkt;
  const std::true_type yes;
  const std::false_type no;
  f32x16_t sa, da, sb, db;
  int kt = 0;
  if constexpr (NSTAGE >= 3) {
    // interior tiles, pipelined: the chains of the next half tile are issued under the arithmetic / dQ products of the current one
    auto step = [&](int kt, auto so_c, auto son_c, auto sreq_c) __attribute__((always_inline)) {
      const int so = so_c, son = son_c, sreq = sreq_c;
      if (wave_active) {
        issue(so, 1, sb, db);
        finish(sa, da, so, kt, 0, no);
      }
      __builtin_amdgcn_sched_barrier(0);   // a half step is the scheduling region: without the fences hipcc hoists the LDS reads of
      land();                              // later half steps to the top of the unrolled triple and spills
      if (kt + 2 < nkt) request(kt + 2, sreq);
      __builtin_amdgcn_sched_barrier(0);
      if (wave_active) {
        issue(son, 0, sa, da);
        finish(sb, db, so, kt, 1, no);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    if (nkt > 0) {
      land();
      if (wave_active) issue(0, 0, sa, da);
    }
    for (; kt + 3 <= n_plain && kt + 3 < nkt; kt += 3) {   // (a tile follows the triple)
      step(kt, CtOff<0>{}, CtOff<STG>{}, CtOff<2 * STG>{});
      step(kt + 1, CtOff<STG>{}, CtOff<2 * STG>{}, CtOff<0>{});
      step(kt + 2, CtOff<2 * STG>{}, CtOff<0>{}, CtOff<STG>{});
    }
    // leftover and masked tiles: general body, one half tile live at a time (its per-score mask code needs the registers the
    // look-ahead would hold; a rolled pipelined loop carries four 16-register accumulators around the back edge and spills,
    // and scratch traffic counts in vmcnt, i.e. drains the DMA ring).  The first of them finds its first half already issued.
    int nreq = kt + 2 < nkt ? kt + 2 : nkt;   // tiles requested so far
    bool pre = nkt > 0;
    for (; kt < nkt; ++kt) {
      const int so = (kt % 3) * STG;
      if (!pre) {
        land();   // tile kt has landed; every wave has left tile kt-1, so the slots of tiles <= kt+2 are free
        for (; nreq < nkt && nreq <= kt + 2; ++nreq) request(nreq, (nreq % 3) * STG);
        if (wave_active) issue(so, 0, sa, da);
      }
      pre = false;
      if (wave_active) {
        finish(sa, da, so, kt, 0, yes);
        if (kt * 64 + 32 < p.Sk) {   // (a ragged last tile of at most 32 keys has an empty second half)
          issue(so, 1, sa, da);
          finish(sa, da, so, kt, 1, yes);
        }
      }
    }
  } else {
    auto step = [&](int kt, auto so_c, auto son_c, auto masked_c) __attribute__((always_inline)) {
      const int so = so_c, son = son_c;
      land();
      if (kt + 1 < nkt) request(kt + 1, son);
      if (wave_active) {
        issue(so, 0, sa, da);
        issue(so, 1, sb, db);
        finish(sa, da, so, kt, 0, masked_c);
        finish(sb, db, so, kt, 1, masked_c);
      }
    };
    for (; kt + 2 <= n_plain; kt += 2) {
      step(kt, CtOff<0>{}, CtOff<STG>{}, no);
      step(kt + 1, CtOff<STG>{}, CtOff<0>{}, no);
    }
    for (; kt < nkt; ++kt) step(kt, RtOff{(kt % 2) * STG}, RtOff{((kt + 1) % 2) * STG}, yes);
  }

  // ---- dQ = scale * dq, out through LDS as whole rows (the ring is free: every DMA has landed and been consumed) ----
  __builtin_amdgcn_s_barrier();
  char* obuf = smem + wid * (32 * OP);
#pragma unroll
  for (int d = 0; d < NDT; ++d)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int col = 32 * d + 8 * c + 4 * hh;
      *(uint2*)(obuf + l31 * OP + col * 2) = make_uint2(pack_bf16x2(dq[d][4 * c] * p.scale, dq[d][4 * c + 1] * p.scale),
                                                        pack_bf16x2(dq[d][4 * c + 2] * p.scale, dq[d][4 * c + 3] * p.scale));
    }
  {
    constexpr int CPR = D / 8;                   // 16-byte chunks per output row
    constexpr int RPI = 64 / CPR;                // rows per store instruction
    const __amdgpu_buffer_rsrc_t rsD = make_rsrc(p.dQ, (unsigned)p.B * p.Sq * p.lddq * 2u);
    const int r_in = lane / CPR, ch = lane % CPR;
    if (lane < RPI * CPR) {
#pragma unroll
      for (int it = 0; it < 32 / RPI + (32 % RPI ? 1 : 0); ++it) {
        const int row = it * RPI + r_in;
        if (row < 32) {
          const i32x4_t v = *(const i32x4_t*)(obuf + row * OP + ch * 16);
          const int qr = q0 + row;
          const int off = (qr < p.Sq) ? ((b * p.Sq + qr) * p.lddq + head * D + ch * 8) * 2 : -1;
          __builtin_amdgcn_raw_buffer_store_b128(v, rsD, off, 0, 0);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// (2) dK/dV kernel: 1-D XCD-aware grid over (key block of 128, head, batch), 4 waves x 32 keys; loops over 64-query tiles.
// ------------------------------------------------------------------------------------------------------
template <int D, bool DROP = false>
__global__ __launch_bounds__(256, (D == 64 ? 2 : 1)) void attn_bwd_dkdv_kernel(AttnBwdArgs p) {
  using I = Img<D>;
  constexpr int NKS = D / 16, NDT = D / 32;
  constexpr int STAGE = 2 * I::TILE + 768;  // Q image, dO image, 64 lse2, 64 delta, 64 dropout row hashes
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int hh = lane >> 5, l31 = lane & 31;
  const int nkb = (p.Sk + 127) / 128;   // 1-D XCD-aware grid: the key blocks of one (batch, head) sweep the same Q / dO
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int bx = tile % nkb, head = (tile / nkb) % p.H, b = tile / (nkb * p.H);
  const int kcol = bx * 128 + wid * 32 + l31;
  const bool kok = kcol < p.Sk;
  const bool wave_active = (int)(bx * 128 + wid * 32) < p.Sk;   // wave-uniform
  const float INF = __builtin_inff();

  const int nqt = (p.Sq + 63) / 64;
  const int qt0 = p.causal ? (bx * 128) / 64 : 0;  // earlier queries see none of these keys

  const __amdgpu_buffer_rsrc_t rsK = make_rsrc(p.K, (unsigned)p.B * p.Sk * p.ldk * 2u);
  const __amdgpu_buffer_rsrc_t rsV = make_rsrc(p.V, (unsigned)p.B * p.Sk * p.ldv * 2u);

  bf16x8_t kf[NKS], vf[NKS];
  {
    const int tok = b * p.Sk + kcol;
    const int ko = (tok * p.ldk + head * D + 8 * hh) * 2, vo = (tok * p.ldv + head * D + 8 * hh) * 2;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      kf[s] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rsK, kok ? ko + s * 32 : -1, 0, 0));
      vf[s] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rsV, kok ? vo + s * 32 : -1, 0, 0));
    }
  }
  const float bias = kok ? (p.kbias ? p.kbias[(size_t)b * p.Sk + kcol] * LOG2E : 0.f) : -INF;

  // Q / dO tiles of 64 queries arrive by LDS-DMA (the K / V scheme of the dQ kernel: 1-KiB pieces, swizzle on the source side, no
  // staging registers and no ds_write pass); the rows' lse2 / delta (/ dropout row hashes) still travel through one register.
  using Cfg = AttnCfg<D>;
  static_assert(Cfg::TILE == I::TILE && Cfg::PITCH == I::PITCH, "one LDS image for both attention backward kernels");
  const unsigned qbytes = (unsigned)p.B * p.Sq * p.ldq * 2u, gbytes = (unsigned)p.B * p.Sq * p.lddo * 2u;
  int q_goff[Cfg::PPW], g_goff[Cfg::PPW];
#pragma unroll
  for (int j = 0; j < Cfg::PPW; ++j) {
    const int row = Cfg::RPP * (wid * Cfg::PPW + j) + lane / Cfg::SLOTS;
    const int ch = (lane % Cfg::SLOTS) ^ Cfg::swz(row);
    q_goff[j] = ch < Cfg::CH ? ((b * p.Sq + row) * p.ldq + head * D + ch * 8) * 2 : -1;
    g_goff[j] = ch < Cfg::CH ? ((b * p.Sq + row) * p.lddo + head * D + ch * 8) * 2 : -1;
  }
  float rstat = 0.f;
  auto gload = [&](int qt, char* stage) {
    attn_dma_tile<D>(p.Q, qbytes, p.dO, gbytes, stage, wid, q_goff, g_goff, qt * 64 * p.ldq * 2, qt * 64 * p.lddo * 2);
    if (tid < 128) {  // threads 0..63: lse2, 64..127: delta
      const int q = qt * 64 + (tid & 63);
      const size_t si = ((size_t)b * p.H + head) * p.Sq + q;
      // the RAW values: nothing may wait for these loads here (a use would drain the DMA pieces requested just above); the
      // -inf -> +inf mapping of an empty row's log-sum-exp happens in lstore
      const float* src = tid < 64 ? p.lse2 : p.delta;
      rstat = (q < p.Sq) ? src[si] : (tid < 64 ? INF : 0.f);
    } else if (DROP && tid < 192) {   // 128..191: the row half of the dropout hash of query q
      const int q = qt * 64 + (tid & 63);
      rstat = __builtin_bit_cast(float, drop_row_hash(p.drop, (unsigned)((b * p.H + head) * p.Sq + q)));
    }
  };
  auto lstore = [&](char* stage) {   // the row statistics; then every DMA piece this wave requested has landed
    if (tid < (DROP ? 192 : 128)) *(float*)(stage + 2 * I::TILE + tid * 4) = (tid < 64 && rstat == -INF) ? INF : rstat;
    wait_vm0();
  };

  const int q4 = (lane >> 2) & 3, p4 = lane & 3, cg = (lane >> 4) & 1;
  int row_addr[NKS], tr_lo[NDT], tr_hi[NDT];  // hoisted LDS addresses (see the dQ kernel)
#pragma unroll
  for (int s = 0; s < NKS; ++s) row_addr[s] = I::off(l31, 2 * s + hh);
#pragma unroll
  for (int d = 0; d < NDT; ++d) {
    const int e = 32 * d + 16 * cg + 4 * p4;
    tr_lo[d] = I::off(4 * hh + q4, e >> 3) + (e & 7) * 2;
    tr_hi[d] = I::off(4 * hh + q4 + 8, e >> 3) + (e & 7) * 2;
  }

  f32x16_t dkt[NDT], dvt[NDT];
#pragma unroll
  for (int d = 0; d < NDT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkt[d][r] = 0.f; dvt[d][r] = 0.f; }

  gload(qt0, smem);   // (qt0 < nqt always: a causal launch has Sq == Sk.)  Unconditional, so that hipcc sees the wait in lstore on
  lstore(smem);       // every path into the loops and stops re-waiting for the K / V fragment loads inside them
  __syncthreads();

  // One tile of 64 queries.  DIAG (some key of the block may exceed some query of the tile: causal launches only) is a
  // compile-time property of the call site, so the non-causal body has no control flow (see the dQ kernel).
  auto tile_body = [&](int qt, auto diag_c, auto bias0_c) {
    constexpr bool DIAG = decltype(diag_c)::value;
    constexpr bool BIAS0 = decltype(bias0_c)::value;   // every key of the block is a real key without a bias: s2 = s*scale - lse
    const int it = qt - qt0;
    const char* cur = smem + (it & 1) * STAGE;
    char* nxt = smem + ((it + 1) & 1) * STAGE;
    const bool more = (qt + 1) < nqt;
    if (more) gload(qt + 1, nxt);
    const float* lsev = (const float*)(cur + 2 * I::TILE);
    const float* delv = lsev + 64;

    const bool h1 = qt * 64 + 32 < p.Sq;   // (a ragged last tile of at most 32 queries has an empty second half: block-uniform)
    if constexpr (BIAS0 && !DIAG && !DROP) {
    if (wave_active && h1) {
      // The vision tower's tiles, software-pipelined AND interleaved (round 4).  A wave issues in order, and an MFMA holds its
      // issue slot until the matrix pipe takes it (32 cycles back to back), so VALU work only runs beside MFMAs of the SAME wave when
      // the two alternate in program order (profiles/r04_attn_dkdv_stamps.txt: chains 460, arithmetic 727, products 536 cycles per
      // half, one after the other).  Here half 1's S / dP chains are issued BETWEEN half 0's exp2 / dS arithmetic (one MFMA, then two score
      // columns = ~8 VALU instructions, the order pinned by sched_barrier fences; fragments and row stats read ahead of the region),
      // and half 0's dV^T / dK^T products between half 1's arithmetic.  Measured: dq + dk/dv at B.H = 384, S = 1025: 464 -> 442 us.
      auto chains = [&](int u, f32x16_t& sa, f32x16_t& dp) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
          const int a = row_addr[s] + 32 * u * I::PITCH;
          const bf16x8_t qfr = *(const bf16x8_t*)(cur + a);
          const bf16x8_t gfr = *(const bf16x8_t*)(cur + I::TILE + a);
          sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[s], sa, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gfr, vf[s], dp, 0, 0, 0);
        }
      };
      auto products = [&](int u, const bf16x8_t (&pf)[2], const bf16x8_t (&dsf)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int s2i = 0; s2i < 2; ++s2i) {
          const int roff = (32 * u + 16 * s2i) * I::PITCH;
#pragma unroll
          for (int d = 0; d < NDT; ++d) {
            const int lo = tr_lo[d] + roff, hi = tr_hi[d] + roff;
            const bf16x8_t gtf = tr_frag3(cur + I::TILE, lo, hi);  // dO^T
            const bf16x8_t qtf = tr_frag3(cur, lo, hi);            // Q^T
            dvt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gtf, pf[s2i], dvt[d], 0, 0, 0);
            dkt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf[s2i], dkt[d], 0, 0, 0);
          }
        }
      };
      f32x16_t sa0, dp0, sa1, dp1;
      bf16x8_t pf0[2], dsf0[2], pf1[2], dsf1[2];
      chains(0, sa0, dp0);
      __builtin_amdgcn_sched_barrier(0);
      // region: half 1's chains beside half 0's arithmetic, the order pinned chunk by chunk (one or two MFMAs, then two score columns)
      auto arith2 = [&](int i, f32x16_t& sa, f32x16_t& dp, const f32x4_t& l4, const f32x4_t& d4) __attribute__((always_inline)) {
        const int c = i >> 1, j = (i & 1) * 2;
        const f32x2_t sc2 = {p.scale_log2, p.scale_log2};
        const f32x2_t s2 = f32x2_t{sa[4 * c + j], sa[4 * c + j + 1]} * sc2 - f32x2_t{l4[j], l4[j + 1]};
        const f32x2_t pr = {__builtin_amdgcn_exp2f(s2[0]), __builtin_amdgcn_exp2f(s2[1])};
        const f32x2_t ds = pr * (f32x2_t{dp[4 * c + j], dp[4 * c + j + 1]} - f32x2_t{d4[j], d4[j + 1]});
        sa[4 * c + j] = pr[0];
        sa[4 * c + j + 1] = pr[1];
        dp[4 * c + j] = ds[0];
        dp[4 * c + j + 1] = ds[1];
      };
      {
        bf16x8_t qf1[NKS], gf1[NKS];
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
          const int a = row_addr[s] + 32 * I::PITCH;
          qf1[s] = *(const bf16x8_t*)(cur + a);
          gf1[s] = *(const bf16x8_t*)(cur + I::TILE + a);
        }
        f32x4_t l4[4], d4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          l4[c] = *(const f32x4_t*)(lsev + 8 * c + 4 * hh);
          d4[c] = *(const f32x4_t*)(delv + 8 * c + 4 * hh);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa1[r] = 0.f; dp1[r] = 0.f; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
          for (int m = i * 2 * NKS / 8; m < (i + 1) * 2 * NKS / 8; ++m) {
            if (m & 1) dp1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gf1[m >> 1], vf[m >> 1], dp1, 0, 0, 0);
            else       sa1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf1[m >> 1], kf[m >> 1], sa1, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          arith2(i, sa0, dp0, l4[i >> 1], d4[i >> 1]);
          if (i & 1) {   // the packed P / dS fragment this column pair completes
            if ((i >> 1) & 1) { pf0[i >> 2] = pack8(sa0, 8 * (i >> 2)); dsf0[i >> 2] = pack8(dp0, 8 * (i >> 2)); }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // region: half 0's products beside half 1's arithmetic
      {
        bf16x8_t gtf[2 * NDT], qtf[2 * NDT];
#pragma unroll
        for (int s2i = 0; s2i < 2; ++s2i) {
#pragma unroll
          for (int d = 0; d < NDT; ++d) {
            const int roff = 16 * s2i * I::PITCH;
            gtf[s2i * NDT + d] = tr_frag3(cur + I::TILE, tr_lo[d] + roff, tr_hi[d] + roff);
            qtf[s2i * NDT + d] = tr_frag3(cur, tr_lo[d] + roff, tr_hi[d] + roff);
          }
        }
        f32x4_t l4[4], d4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          l4[c] = *(const f32x4_t*)(lsev + 32 + 8 * c + 4 * hh);
          d4[c] = *(const f32x4_t*)(delv + 32 + 8 * c + 4 * hh);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
          for (int m = i * 4 * NDT / 8; m < (i + 1) * 4 * NDT / 8; ++m) {
            const int f = m >> 1, s2i = f / NDT, d = f % NDT;
            if (m & 1) dkt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf[f], dsf0[s2i], dkt[d], 0, 0, 0);
            else       dvt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gtf[f], pf0[s2i], dvt[d], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          arith2(i, sa1, dp1, l4[i >> 1], d4[i >> 1]);
          if ((i & 3) == 3) { pf1[i >> 2] = pack8(sa1, 8 * (i >> 2)); dsf1[i >> 2] = pack8(dp1, 8 * (i >> 2)); }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      products(1, pf1, dsf1);
    }
    }
    const bool pipelined = BIAS0 && !DIAG && !DROP && h1;
    if (wave_active && !pipelined) {   // waves past Sk (ragged last block) only help staging
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && qt * 64 + 32 >= p.Sq) continue;   // a ragged last tile of at most 32 queries: empty second half (block-uniform)
      f32x16_t sa, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sa[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        const int a = row_addr[s] + 32 * u * I::PITCH;
        const bf16x8_t qfr = *(const bf16x8_t*)(cur + a);
        const bf16x8_t gfr = *(const bf16x8_t*)(cur + I::TILE + a);
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[s], sa, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gfr, vf[s], dp, 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4_t l4 = *(const f32x4_t*)(lsev + 32 * u + 8 * c + 4 * hh);
        const f32x4_t d4 = *(const f32x4_t*)(delv + 32 * u + 8 * c + 4 * hh);
        float keep4[4] = {1.f, 1.f, 1.f, 1.f};
        if constexpr (DROP) {
          const i32x4_t rh4 = *(const i32x4_t*)(delv + 64 + 32 * u + 8 * c + 4 * hh);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            keep4[j] = drop_keep(p.drop, (unsigned)rh4[j], (unsigned)kcol) ? p.drop.inv_keep : 0.f;
            dp[4 * c + j] *= keep4[j];
          }
        }
        if constexpr (BIAS0 && !DIAG && !DROP) {   // the vision tower's body: packed fma / sub / mul, two scores per instruction
          const f32x2_t sc2 = {p.scale_log2, p.scale_log2};
#pragma unroll
          for (int j = 0; j < 4; j += 2) {
            const f32x2_t s2 = f32x2_t{sa[4 * c + j], sa[4 * c + j + 1]} * sc2 - f32x2_t{l4[j], l4[j + 1]};
            const f32x2_t pr = {__builtin_amdgcn_exp2f(s2[0]), __builtin_amdgcn_exp2f(s2[1])};
            const f32x2_t ds = pr * (f32x2_t{dp[4 * c + j], dp[4 * c + j + 1]} - f32x2_t{d4[j], d4[j + 1]});
            sa[4 * c + j] = pr[0];
            sa[4 * c + j + 1] = pr[1];
            dp[4 * c + j] = ds[0];
            dp[4 * c + j + 1] = ds[1];
          }
        } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float s2 = BIAS0 ? __builtin_fmaf(sa[4 * c + j], p.scale_log2, -l4[j]) : __builtin_fmaf(sa[4 * c + j], p.scale_log2, bias) - l4[j];
          if constexpr (DIAG) {
            const int q = qt * 64 + 32 * u + 8 * c + 4 * hh + j;
            if (kcol > q) s2 = -INF;
          }
          const float pr = __builtin_amdgcn_exp2f(s2);
          sa[4 * c + j] = DROP ? pr * keep4[j] : pr;     // dV takes the dropped probabilities
          dp[4 * c + j] = pr * (dp[4 * c + j] - d4[j]);
        }
        }
      }
#pragma unroll
      for (int s2i = 0; s2i < 2; ++s2i) {
        const bf16x8_t pf = pack8(sa, 8 * s2i);
        const bf16x8_t dsf = pack8(dp, 8 * s2i);
        const int roff = (32 * u + 16 * s2i) * I::PITCH;
#pragma unroll
        for (int d = 0; d < NDT; ++d) {
          const int lo = tr_lo[d] + roff, hi = tr_hi[d] + roff;
          const bf16x8_t gtf = tr_frag3(cur + I::TILE, lo, hi);  // dO^T
          const bf16x8_t qtf = tr_frag3(cur, lo, hi);            // Q^T
          dvt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gtf, pf, dvt[d], 0, 0, 0);
          dkt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf, dkt[d], 0, 0, 0);
        }
      }
    }
    }  // wave_active
    if (more) lstore(nxt);
    else wait_vm0();   // (keeps hipcc's vmcnt bookkeeping clean on the loop's back edge: see common.h)
    __syncthreads();
  };
  // causal launches: the first tiles (queries up to the block's last key) straddle the diagonal, the rest lie below it
  int n_diag_end = qt0;
  if (p.causal) { n_diag_end = (bx * 128 + 127) / 64 + 1; if (n_diag_end > nqt) n_diag_end = nqt; }
  const bool bias0 = p.kbias == nullptr && bx * 128 + 128 <= p.Sk;   // block-uniform
  if (bias0) {
    for (int qt = qt0; qt < n_diag_end; ++qt) tile_body(qt, std::true_type{}, std::true_type{});
    for (int qt = n_diag_end; qt < nqt; ++qt) tile_body(qt, std::false_type{}, std::true_type{});
  } else {
    for (int qt = qt0; qt < n_diag_end; ++qt) tile_body(qt, std::true_type{}, std::false_type{});
    for (int qt = n_diag_end; qt < nqt; ++qt) tile_body(qt, std::false_type{}, std::false_type{});
  }

  if (kok) {
    bf16_t* krow = p.dK + (size_t)(b * p.Sk + kcol) * p.lddk + head * D;
    bf16_t* vrow = p.dV + (size_t)(b * p.Sk + kcol) * p.lddv + head * D;
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int col = 32 * d + 8 * c + 4 * hh;
        uint2 pk = make_uint2(pack_bf16x2(dkt[d][4 * c] * p.scale, dkt[d][4 * c + 1] * p.scale),
                              pack_bf16x2(dkt[d][4 * c + 2] * p.scale, dkt[d][4 * c + 3] * p.scale));
        *reinterpret_cast<uint2*>(krow + col) = pk;
        uint2 pv = make_uint2(pack_bf16x2(dvt[d][4 * c], dvt[d][4 * c + 1]),
                              pack_bf16x2(dvt[d][4 * c + 2], dvt[d][4 * c + 3]));
        *reinterpret_cast<uint2*>(vrow + col) = pv;
      }
  }
}

template <int D, bool DROP = false>
int launch_attn_bwd(const AttnBwdArgs& a, hipStream_t stream) {
  using I = Img<D>;
  constexpr int LDS_KV = 2 * (2 * I::TILE + 768);
  auto k1b = attn_bwd_dq2_kernel<D, DROP>;
  auto k2 = attn_bwd_dkdv_kernel<D, DROP>;
  using Cfg = AttnCfg<D>;
  constexpr int OBUF = 4 * 32 * (2 * D + 16);
  constexpr int LDS_DQ2 = Cfg::NSTAGE * Cfg::STAGE > OBUF ? Cfg::NSTAGE * Cfg::STAGE : OBUF;
  static DevOnce attr_set;
  if (attr_set.need()) {
    if (hipFuncSetAttribute((const void*)k1b, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DQ2) != hipSuccess ||
        hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_KV) != hipSuccess)
      return LC2IS_ERR_LAUNCH;
    attr_set.done();
  }
  hipLaunchKernelGGL(k1b, dim3(((a.Sq + 127) / 128) * a.H * a.B), dim3(256), LDS_DQ2, stream, a);
  int rc = lc2is_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(k2, dim3(((a.Sk + 127) / 128) * a.H * a.B), dim3(256), LDS_KV, stream, a);
  return lc2is_check_launch();
}

}  // namespace

static int attention_bwd_impl(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                              const void* O, int ldo, const void* dO, int lddo, void* dQ, int lddq,
                              void* dK, int lddk, void* dV, int lddv, const float* lse2, float* delta,
                              const float* kbias, int B, int H, int Sq, int Sk, int D, float scale,
                              int causal, float p_drop, unsigned long long seed, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!Q || !K || !V || !O || !dO || !dQ || !dK || !dV || !lse2 || !delta) return LC2IS_ERR_NULL;
  if (B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0) return LC2IS_ERR_SHAPE;
  const int hd = H * D;
  if (ldq < hd || ldk < hd || ldv < hd || ldo < hd || lddo < hd || lddq < hd || lddk < hd || lddv < hd)
    return LC2IS_ERR_SHAPE;
  if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 8 || lddo % 8 || lddq % 4 || lddk % 4 || lddv % 4)
    return LC2IS_ERR_SHAPE;
  if (causal && Sq != Sk) return LC2IS_ERR_UNSUPPORTED;
  if (!(p_drop >= 0.f && p_drop < 1.f)) return LC2IS_ERR_UNSUPPORTED;
  const double lim = 2147483648.0;
  if ((double)B * (Sq + 64) * ldq * 2.0 >= lim || (double)B * (Sq + 64) * lddo * 2.0 >= lim ||
      (double)B * (Sq + 64) * ldo * 2.0 >= lim || (double)B * (Sk + 64) * ldk * 2.0 >= lim ||
      (double)B * (Sk + 64) * ldv * 2.0 >= lim)
    return LC2IS_ERR_UNSUPPORTED;
  AttnBwdArgs a{(const bf16_t*)Q, ldq, (const bf16_t*)K, ldk, (const bf16_t*)V, ldv, (const bf16_t*)O, ldo,
                (const bf16_t*)dO, lddo, (bf16_t*)dQ, lddq, (bf16_t*)dK, lddk, (bf16_t*)dV, lddv, lse2, delta,
                kbias, B, H, Sq, Sk, scale, scale * LOG2E, causal, make_drop_cfg(p_drop, seed)};
  if (a.drop.thr) {
    if ((double)B * H * Sq >= 4294967296.0) return LC2IS_ERR_UNSUPPORTED;
    switch (D) {
      case 64: return launch_attn_bwd<64, true>(a, stream);
      case 96: return launch_attn_bwd<96, true>(a, stream);
      case 128: return launch_attn_bwd<128, true>(a, stream);
      default: return LC2IS_ERR_UNSUPPORTED;
    }
  }
  switch (D) {
    case 64: return launch_attn_bwd<64>(a, stream);
    case 96: return launch_attn_bwd<96>(a, stream);
    case 128: return launch_attn_bwd<128>(a, stream);
    default: return LC2IS_ERR_UNSUPPORTED;
  }
}

extern "C" int lc2is_attention_bwd(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                                   const void* O, int ldo, const void* dO, int lddo, void* dQ, int lddq,
                                   void* dK, int lddk, void* dV, int lddv, const float* lse2, float* delta,
                                   const float* kbias, int B, int H, int Sq, int Sk, int D, float scale,
                                   int causal, lc2is_stream_t stream) {
  return attention_bwd_impl(Q, ldq, K, ldk, V, ldv, O, ldo, dO, lddo, dQ, lddq, dK, lddk, dV, lddv, lse2, delta, kbias, B, H,
                            Sq, Sk, D, scale, causal, 0.f, 0ULL, stream);
}

extern "C" int lc2is_attention_bwd_dropout(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                                           const void* O, int ldo, const void* dO, int lddo, void* dQ, int lddq,
                                           void* dK, int lddk, void* dV, int lddv, const float* lse2, float* delta,
                                           const float* kbias, int B, int H, int Sq, int Sk, int D, float scale,
                                           int causal, float p_drop, unsigned long long seed, lc2is_stream_t stream) {
  return attention_bwd_impl(Q, ldq, K, ldk, V, ldv, O, ldo, dO, lddo, dQ, lddq, dK, lddk, dV, lddv, lse2, delta, kbias, B, H,
                            Sq, Sk, D, scale, causal, p_drop, seed, stream);
}
