// Fused multi-head attention forward (flash-style, online softmax) for gfx950.
// replaces: hf eager_attention_forward / sdpa in CLIPAttention.forward (hf:modeling_clip.py:259-335) for
//   the ViT (S=1025, 12x64) and text (causal ∧ padding, 8x64) encoders, and torch
//   F.multi_head_attention_forward's softmax(QK^T/sqrt(d))V (torch:nn/functional.py:6206) for the decoder's
//   self-attention (8x96, S=1024) and text cross-attention with memory_key_padding_mask (model/decoder.py:20).
//
// Layout: Q/K/V are read in place from the projection GEMM outputs — token-major rows (b*S + s), head h in
// columns [h*D, (h+1)*D) with an arbitrary row stride (so the packed [M, 3*H*D] QKV buffer is never
// permuted); O is written token-major [B*Sq, H*D], directly consumable by the out-projection GEMM.
//
// Work split: block = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries.  Scores are
// computed TRANSPOSED, S^T[key][query] = K·Q^T with 32x32x16 bf16 MFMAs, so a lane holds 16 keys of ONE
// query per 32-key half tile: the softmax max / sum are per-lane loops (the two lane halves of a row only talk when
// the row's frame moves, see below), and the fp32 score registers, packed pairwise to bf16, are already the B operand
// of O^T[d][query] += V^T·P^T (accumulator-as-operand, no LDS round trip).  V^T fragments come from the row-major V
// tile with ds_read_b64_tr_b16.  K/V tiles of 64 keys arrive by LDS-DMA (buffer_load ... lds, range-checked, no staging
// registers) into a ring of swizzled images (attn_common.h) that both kinds of read take without bank conflicts.
// Masks (key tail, key padding, causal) are applied per score, only in tiles that need one.
#include "attn_common.h"
#include "lc2is_hip.h"
#include <cstdlib>

namespace {

struct AttnFwdArgs {
  const bf16_t* Q; int ldq;
  const bf16_t* K; int ldk;
  const bf16_t* V; int ldv;
  bf16_t* O; int ldo;
  float* lse2;          // [B,H,Sq], log2-domain logsumexp of the scaled scores (for backward), may be null
  const float* kbias;   // [B,Sk] additive key bias in natural-log units (0 / -inf), may be null
  int B, H, Sq, Sk;
  float scale_log2;     // softmax scale * log2(e)
  int causal;
  DropCfg drop;         // attention-probability dropout (DROP instantiations only)
};

// Structure.  Measured on MI355X at S = 1025, D = 64 (rocprofv3 SQ counters — tools/pmc_attn.sh —, s_memtime phase stamps of a diagnostic build
// and key-length sweeps — tools/attn_fixedcost.py): the launch costs 20 us + 8.25 us per 64-key tile, and a wave of the
// un-pipelined form spends 3300 cycles per tile of which 2070 sit in ONE serial chain
//     ds_read K -> 4 dependent MFMAs -> scale -> max -> branch -> exp2 -> pack -> ds_read V^T -> P.V
// that hipcc cannot overlap because the frame check splits it into basic blocks; three such waves keep the matrix pipe 47 % busy.
//  * software pipeline over 32-key HALF tiles: the S^T chain of half h+1 (LDS reads + MFMAs) is issued in the same basic block
//    as exp2 / row sums / pack / V^T reads / P.V of half h, so its latency runs under the VALU work and vice versa; the only
//    branch of a half step — the frame check — sits at the block boundary in front of it;
//  * K/V tiles arrive by LDS-DMA into a ring (3 deep at D = 64) behind counted s_waitcnt vmcnt + one raw s_barrier per tile,
//    placed in the middle of a tile's two half steps (where the next tile's first S^T chain needs its data);
//  * every row keeps its state (O, l) in a FRAME m_ref: O = sum_k exp2(s_k - m_ref) v_k.  A half tile adds exp2(s - m_ref) in the
//    same frame, so nothing is rescaled and the lane halves that share a row do not talk to each other.  Only when some row's
//    half-tile maximum exceeds its frame by more than FRAME_THR (or a row meets its first unmasked key) the wave takes the rare
//    branch: the halves exchange their maxima, the row moves to the frame of its new maximum (O, l times exp2(m_ref - m_ref'),
//    the waiting scores shifted by the same amount).  Probabilities are <= 2^FRAME_THR and a row's frame is always the
//    maximum of real scores: no overflow, no row-wide underflow;
//  * a score costs one fma (scale and frame), half a max3, one exp2, one add and half a pack (tried and dropped: Q pre-multiplied
//    by scale*log2(e) in bf16 with -m_ref fed through an extra MFMA saves the fma but rounds Q once more — 2-3e-3 on scores and
//    log-sum-exp — and measured no faster);
//  * O leaves through LDS as whole 2D-byte rows (row-per-lane stores touch 32 cache lines per instruction).
// DROP: dropout on the attention probabilities (torch multi_head_attention_forward dropout_p in training mode,
// torch:nn/functional.py:6206): l sums the UNDROPPED probabilities, O accumulates keep * P / (1 - p);
// coordinate of a score = (row (b*H + h)*Sq + q, column key), so the backward kernels regenerate the same decisions.
constexpr float FRAME_THR = 6.0f;
// (round 5, tried and removed: the per-score fma and the row-sum adds pinned to single v_fma_f32 / v_add_f32 through inline asm
//  instead of hipcc's v_pk_fma_f32 / v_pk_add_f32 — 137 vs 139 us at the ViT shape, 124 vs 120 at the decoder's, nothing in the step;
//  and an asm VALU instruction that is the FIRST reader of an MFMA result gets no wait states from hipcc: the D = 96 instantiation
//  computed garbage.  profiles/r05_attn_fwd_scalar_valu_ab.txt)

// NQ = 32-query groups per wave.  NQ = 1: 3 waves/SIMD at D = 64.  NQ = 2 (64 queries per wave, 256 per block): every K row
// fragment and every V^T fragment read from LDS feeds TWO MFMA chains (half the LDS bytes per MFMA, and the S^T chains of the two
// groups are independent, so they issue back to back instead of waiting out each other's latency) at 2 waves/SIMD.
template <int D, bool DROP, int NQ>
__global__ __launch_bounds__(256, NQ == 2 ? (D == 64 ? 2 : 1) : ((D == 64) ? 3 : 2)) void attn_fwd_kernel(AttnFwdArgs p) {
  using Cfg = AttnCfg<D>;
  constexpr int PITCH = Cfg::PITCH, NSTAGE = Cfg::NSTAGE, PD = NSTAGE - 1;
  constexpr int NKS = D / 16, NDT = D / 32;
  constexpr int QB = 128 * NQ;          // queries per block (4 waves x 32 NQ)
  constexpr int OP = 2 * D + 16;        // row pitch of the output staging image (bytes)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, l31 = lane & 31;
  // 1-D grid, XCD-aware: the query blocks of one (batch, head) — which stream the same K / V — are neighbours in the tile
  // order, and xcd_remap gives every XCD (its own L2) one contiguous chunk of that order
  const int nqb = (p.Sq + QB - 1) / QB;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int bx = tile % nqb, head = (tile / nqb) % p.H, b = tile / (nqb * p.H);
  const int q0 = bx * QB + wid * 32 * NQ;             // first query of this wave
  int qrow[NQ];
#pragma unroll
  for (int g = 0; g < NQ; ++g) qrow[g] = q0 + 32 * g + l31;
  const bool wave_active = q0 < p.Sq;                 // wave-uniform; waves wholly past Sq only help staging
  const float NEG_INF = -__builtin_inff();

  int nkt = (p.Sk + 63) / 64;
  if (p.causal) {
    const int lim = (bx * QB + QB + 63) / 64;  // keys <= last query of the block
    if (lim < nkt) nkt = lim;
  }

  const unsigned qbytes = (unsigned)p.B * p.Sq * p.ldq * 2u;
  const unsigned kbytes = (unsigned)p.B * p.Sk * p.ldk * 2u, vbytes = (unsigned)p.B * p.Sk * p.ldv * 2u;

  // LDS-DMA bookkeeping: piece j of this wave covers rows RPP*(wid*PPW + j) .. +RPP-1 of the tile; lane l fills slot l % SLOTS
  // of row l / SLOTS with the source chunk that the swizzle maps there (chunks past the row's data read as zero: offset -1)
  int k_goff[Cfg::PPW], v_goff[Cfg::PPW];
#pragma unroll
  for (int j = 0; j < Cfg::PPW; ++j) {
    const int row = Cfg::RPP * (wid * Cfg::PPW + j) + lane / Cfg::SLOTS;
    const int ch = (lane % Cfg::SLOTS) ^ Cfg::swz(row);
    k_goff[j] = ch < Cfg::CH ? ((b * p.Sk + row) * p.ldk + head * D + ch * 8) * 2 : -1;
    v_goff[j] = ch < Cfg::CH ? ((b * p.Sk + row) * p.ldv + head * D + ch * 8) * 2 : -1;
  }
  auto request = [&](int kt, int so) __attribute__((always_inline)) {   // tile kt -> the ring slot at byte offset so = (kt % NSTAGE) * STAGE
    attn_dma_tile<D>(p.K, kbytes, p.V, vbytes, smem + so, wid, k_goff, v_goff, kt * 64 * p.ldk * 2, kt * 64 * p.ldv * 2);
  };
  // the first PD tiles are on their way before anything else happens (their latency overlaps the Q loads below)
#pragma unroll
  for (int i = 0; i < PD; ++i)
    if (i < nkt) request(i, i * Cfg::STAGE);

  // Q fragments (B operand of S^T = K·Q^T): lane holds Q[qrow][16s + 8hh .. +7]
  bf16x8_t qf[NQ][NKS];
  unsigned drop_rh[NQ];
#pragma unroll
  for (int g = 0; g < NQ; ++g) {
    const __amdgpu_buffer_rsrc_t rsQ = make_rsrc(p.Q, qbytes);
    const int off = (qrow[g] < p.Sq) ? ((b * p.Sq + qrow[g]) * p.ldq + head * D + 8 * hh) * 2 : -1;
#pragma unroll
    for (int s = 0; s < NKS; ++s)
      qf[g][s] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rsQ, off < 0 ? -1 : off + s * 32, 0, 0));
    drop_rh[g] = DROP ? drop_row_hash(p.drop, (unsigned)((b * p.H + head) * p.Sq + qrow[g])) : 0u;
  }

  // fragment addresses in a stage (the swizzle only looks at row bits 0..3: 16- / 32-row steps are immediates on these bases)
  const int q4 = (lane >> 2) & 3, p4 = lane & 3, cg = (lane >> 4) & 1;
  const unsigned smem_a = (unsigned)(size_t)LDS_PTR(smem);
  unsigned k_row[NKS], v_lo[NDT], v_hi[NDT];   // LDS addresses in stage 0
#pragma unroll
  for (int s = 0; s < NKS; ++s) {
    k_row[s] = smem_a + Cfg::off(l31, 2 * s + hh);          // + 32t*PITCH
    asm volatile("" : "+v"(k_row[s]));
  }
#pragma unroll
  for (int d = 0; d < NDT; ++d) {
    const int e = 32 * d + 16 * cg + 4 * p4;
    v_lo[d] = smem_a + Cfg::TILE + Cfg::off(4 * hh + q4, e >> 3) + (e & 7) * 2;        // + (32t + 16 s2)*PITCH
    v_hi[d] = smem_a + Cfg::TILE + Cfg::off(4 * hh + q4 + 8, e >> 3) + (e & 7) * 2;
    asm volatile("" : "+v"(v_lo[d]), "+v"(v_hi[d]));
  }

  f32x16_t ot[NQ][NDT];
  f32x2_t lsum2[NQ];               // this lane's share of the row sum (its 16 keys per half tile), even / odd scores apart (packed adds)
  float m_ref[NQ];                 // the row's frame
  bool counted[NQ];                // has the row met an unmasked key (its frame is then a real score's maximum)
#pragma unroll
  for (int g = 0; g < NQ; ++g) {
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[g][d][r] = 0.f;
    lsum2[g] = f32x2_t{0.f, 0.f};
    m_ref[g] = 0.f;
    counted[g] = false;
  }

  // S^T = K · Q^T of one 32-key half tile (raw, unscaled): 4 (D/16) LDS row reads + dependent MFMAs
  auto issue_s = [&](int so, int t, f32x16_t (&st)[NQ]) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < NQ; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[g][r] = 0.f;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      const bf16x8_t kf = lds_read_b128(k_row[s] + (unsigned)(so + 32 * t * PITCH));
#pragma unroll
      for (int g = 0; g < NQ; ++g) st[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[g][s], st[g], 0, 0, 0);
    }
  };
  // block 1 of a half step: scale into the row's frame, masks, per-lane maximum, frame check (the only branch)
  // (masked_c = false_type: the caller knows the tile has no key tail, no causal edge and no key bias — no mask code at all)
  auto frame1 = [&](f32x16_t& st, int g, int kt, int t, auto masked_c) __attribute__((always_inline)) {
    const bool tail = (kt * 64 + 64 > p.Sk);
    const bool diag = p.causal && (kt * 64 + 63 > bx * QB);  // some key may exceed some query
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = __builtin_fmaf(st[r], p.scale_log2, -m_ref[g]);
    if (decltype(masked_c)::value && (tail || diag || p.kbias != nullptr)) {   // key tail / key padding / causal edge: -inf (or the additive key bias) per score
      int key0 = kt * 64 + 32 * t + 4 * hh, qr = qrow[g];
      asm volatile("" : "+v"(key0), "+v"(qr));   // keeps the per-score compares INSIDE this branch (hipcc hoists them otherwise)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int key = key0 + 8 * c + j;
          float sc = st[4 * c + j];
          if (key >= p.Sk) sc = NEG_INF;                                                       // key tail (the tile's zero rows)
          else if (p.kbias) sc += p.kbias[(size_t)b * p.Sk + key] * 1.44269504088896341f;      // 0 / -inf key padding
          if (diag && key > qr) sc = NEG_INF;
          st[4 * c + j] = sc;
        }
    }
    float ma = max3f(st[0], st[1], st[2]), mb = max3f(st[3], st[4], st[5]);
    ma = max3f(ma, st[6], st[7]);
    mb = max3f(mb, st[8], st[9]);
    ma = max3f(ma, st[10], st[11]);
    mb = max3f(mb, st[12], st[13]);
    const float mx = max2f(max3f(ma, st[14], st[15]), mb);
    if (__any((mx > FRAME_THR) || (!counted[g] && mx > NEG_INF))) {
      // rare: some row outgrew its frame (or met its first key): move those rows to the frame of their new maximum
      const float mxx = fmaxf(mx, __shfl_xor(mx, 32, 64));            // the row's maximum over both lane halves
      const bool move = (mxx > FRAME_THR) || (!counted[g] && mxx > NEG_INF);
      const float delta = move ? mxx : 0.f;
      // 1 where nothing moves.  A row's FIRST frame is not a rescale: O and l are exactly 0 and the dummy frame 0 means nothing,
      // so exp2(-delta) must not be formed there (a first maximum below -128 gives +inf, and 0 * inf = NaN in O and l)
      const float corr = counted[g] ? __builtin_amdgcn_exp2f(-delta) : 1.f;
#pragma unroll
      for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[g][d][r] *= corr;
      lsum2[g] *= corr;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] -= delta;
      m_ref[g] += delta;
      counted[g] = counted[g] || (mxx > NEG_INF);
    }
  };
  auto frame = [&](f32x16_t (&st)[NQ], int kt, int t, auto masked_c) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < NQ; ++g) frame1(st[g], g, kt, t, masked_c);
  };
  // block 2 of a half step: P = exp2(s - m_ref), row sums, pack, O^T += V^T · P^T   (+ the next half's S^T chain, see the loop)
  auto finish = [&](f32x16_t (&st)[NQ], int so, int kt, int t) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < NQ; ++g) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[g][r] = __builtin_amdgcn_exp2f(st[g][r]);
#pragma unroll
      for (int r = 0; r < 8; ++r) lsum2[g] += f32x2_t{st[g][2 * r], st[g][2 * r + 1]};   // v_pk_add_f32
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8_t pf[NQ];
#pragma unroll
      for (int g = 0; g < NQ; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = 8 * s2 + j;
          float pv = st[g][r];
          if constexpr (DROP) {
            const unsigned key = (unsigned)(kt * 64 + 32 * t + 8 * (r >> 2) + 4 * hh + (r & 3));
            pv = drop_keep(p.drop, drop_rh[g], key) ? pv * p.drop.inv_keep : 0.f;
          }
          pf[g][j] = (__bf16)pv;
        }
#pragma unroll
      for (int d = 0; d < NDT; ++d) {
        const int roff = so + (32 * t + 16 * s2) * PITCH;
        const bf16x8_t vf = tr_frag2a(v_lo[d] + (unsigned)roff, v_hi[d] + (unsigned)roff);
#pragma unroll
        for (int g = 0; g < NQ; ++g) ot[g][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[g], ot[g][d], 0, 0, 0);
      }
    }
  };
  auto land = [&]() __attribute__((always_inline)) {   // this wave's outstanding DMAs have landed, and behind the barrier so have every wave's
    wait_vm0();
    __builtin_amdgcn_s_barrier();
  };

  f32x16_t sa[NQ], sb[NQ];   // raw scores of the half in progress / the half issued ahead
  // Leading tiles that need no mask code: whole inside Sk, left of the causal edge, no key bias.  They run in a main loop
  // unrolled over the ring (every stage offset is then a compile-time constant: the 12 swizzled fragment bases of a lane take
  // immediate offsets instead of two VALU adds per LDS read — 52 of the 226 VALU instructions of a wave-tile, in a loop that
  // is VALU-issue bound) and instantiated without the mask branch; the leftover and masked tiles take the rolled loop.
  int n_plain = p.kbias ? 0 : p.Sk / 64;
  if (p.causal && (QB / 64) * bx < n_plain) n_plain = (QB / 64) * bx;
  if (nkt < n_plain) n_plain = nkt;
  constexpr int STG = Cfg::STAGE;
  int kt = 0;
  if constexpr (Cfg::NSTAGE >= 3) {
    // ring of three: tile kt+1 is awaited in the MIDDLE of tile kt (its first S^T chain is issued under the second half of tile
    // kt) and tile kt+2 is requested there, into the slot of tile kt-1 which every wave has left when it reaches that barrier
    auto step = [&](int kt, auto so_c, auto son_c, auto sreq_c, auto masked_c, auto more_c) __attribute__((always_inline)) {
      const int so = so_c, son = son_c, sreq = sreq_c;
      const bool more = decltype(more_c)::value || kt + 1 < nkt;
      // a ragged last tile with at most 32 keys (S = 64 n + 1): its second half holds nothing (wave-uniform; never on interior tiles)
      const bool h1 = !decltype(masked_c)::value || kt * 64 + 32 < p.Sk;
      if (wave_active) {
        frame(sa, kt, 0, masked_c);
        if (h1) issue_s(so, 1, sb);           // second half of this tile, under the exp2 / P.V of the first
        finish(sa, so, kt, 0);
      }
      if (more) {
        land();
        if (kt + 2 < nkt) request(kt + 2, sreq);
      }
      if (wave_active && h1) {
        frame(sb, kt, 1, masked_c);
        if (more) issue_s(son, 0, sa);        // first half of the next tile, under the exp2 / P.V of this one
        finish(sb, so, kt, 1);
      }
    };
    if (nkt > 0) {
      land();
      if (wave_active) issue_s(0, 0, sa);
    }
    const std::true_type yes;
    const std::false_type no;
    for (; kt + 3 <= n_plain && kt + 3 < nkt; kt += 3) {   // (a tile follows the triple: `more` is known)
      step(kt, CtOff<0>{}, CtOff<STG>{}, CtOff<2 * STG>{}, no, yes);
      step(kt + 1, CtOff<STG>{}, CtOff<2 * STG>{}, CtOff<0>{}, no, yes);
      step(kt + 2, CtOff<2 * STG>{}, CtOff<0>{}, CtOff<STG>{}, no, yes);
    }
    for (; kt < nkt; ++kt)
      step(kt, RtOff{(kt % 3) * STG}, RtOff{((kt + 1) % 3) * STG}, RtOff{((kt + 2) % 3) * STG}, yes, no);
  } else {
    // ring of two (32-KiB stages at D >= 96): one barrier at the top of a tile, the next tile requested behind it
    auto step = [&](int kt, auto so_c, auto son_c, auto masked_c) __attribute__((always_inline)) {
      const int so = so_c, son = son_c;
      land();
      if (kt + 1 < nkt) request(kt + 1, son);
      const bool h1 = !decltype(masked_c)::value || kt * 64 + 32 < p.Sk;
      if (wave_active) {
        issue_s(so, 0, sa);
        frame(sa, kt, 0, masked_c);
        if (h1) issue_s(so, 1, sb);
        finish(sa, so, kt, 0);
        if (h1) {
          frame(sb, kt, 1, masked_c);
          finish(sb, so, kt, 1);
        }
      }
    };
    const std::true_type yes;
    const std::false_type no;
    for (; kt + 2 <= n_plain; kt += 2) {
      step(kt, CtOff<0>{}, CtOff<STG>{}, no);
      step(kt + 1, CtOff<STG>{}, CtOff<0>{}, no);
    }
    for (; kt < nkt; ++kt) step(kt, RtOff{(kt % 2) * STG}, RtOff{((kt + 1) % 2) * STG}, yes);
  }

  // ---- normalise; O goes out through LDS as whole rows (the ring is free: every DMA has landed and been consumed) ----
  __builtin_amdgcn_s_barrier();   // all waves are past their last reads of the ring
  char* obuf = smem + wid * (32 * NQ * OP);
#pragma unroll
  for (int g = 0; g < NQ; ++g) {
    const float lsum = lsum2[g][0] + lsum2[g][1];
    const float l_tot = lsum + __shfl_xor(lsum, 32, 64);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    if (p.lse2 && hh == 0 && qrow[g] < p.Sq)
      p.lse2[((size_t)b * p.H + head) * p.Sq + qrow[g]] = l_tot > 0.f ? m_ref[g] + __builtin_amdgcn_logf(l_tot) : NEG_INF;
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int col = 32 * d + 8 * c + 4 * hh;
        *(uint2*)(obuf + (32 * g + l31) * OP + col * 2) = make_uint2(pack_bf16x2(ot[g][d][4 * c] * inv, ot[g][d][4 * c + 1] * inv),
                                                                     pack_bf16x2(ot[g][d][4 * c + 2] * inv, ot[g][d][4 * c + 3] * inv));
      }
  }
  // (each wave reads back only what it wrote: no barrier, the compiler's lgkmcnt wait orders the LDS accesses of one wave)
  {
    constexpr int CPR = D / 8;                   // 16-byte chunks per output row
    constexpr int RPI = 64 / CPR;                // rows per store instruction (lanes of one row are contiguous: a whole row segment)
    const __amdgpu_buffer_rsrc_t rsO = make_rsrc(p.O, (unsigned)p.B * p.Sq * p.ldo * 2u);
    const int r_in = lane / CPR, ch = lane % CPR;
    if (lane < RPI * CPR) {
#pragma unroll
      for (int it = 0; it < 32 * NQ / RPI + (32 * NQ % RPI ? 1 : 0); ++it) {
        const int row = it * RPI + r_in;
        if (row < 32 * NQ) {
          const i32x4_t v = *(const i32x4_t*)(obuf + row * OP + ch * 16);
          const int qr = q0 + row;
          const int off = (qr < p.Sq) ? ((b * p.Sq + qr) * p.ldo + head * D + ch * 8) * 2 : -1;
          __builtin_amdgcn_raw_buffer_store_b128(v, rsO, off, 0, 0);
        }
      }
    }
  }
}

template <int D, bool DROP, int NQ>
int launch_attn_fwd_nq(const AttnFwdArgs& a, hipStream_t stream) {
  using Cfg = AttnCfg<D>;
  auto kern = attn_fwd_kernel<D, DROP, NQ>;
  constexpr int OBUF = 4 * 32 * NQ * (2 * D + 16);
  constexpr int LDS = Cfg::NSTAGE * Cfg::STAGE > OBUF ? Cfg::NSTAGE * Cfg::STAGE : OBUF;
  static DevOnce attr_set;
  if (attr_set.need()) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return LC2IS_ERR_LAUNCH;
    attr_set.done();
  }
  hipLaunchKernelGGL(kern, dim3(((a.Sq + 128 * NQ - 1) / (128 * NQ)) * a.H * a.B), dim3(256), LDS, stream, a);
  return lc2is_check_launch();
}

// NQ = 1 (32 queries per wave) is the one instantiation the library carries.  NQ = 2 (64 queries per wave: built and parity-green in
// round 4, 161 vs 156 us at the ViT shape, 2 waves/SIMD instead of 3 — tools/probes/README.md, profiles/r04_attn_fwd_nq_ab.txt)
// stays a template parameter of the kernel source only.
template <int D, bool DROP>
int launch_attn_fwd(const AttnFwdArgs& a, hipStream_t stream) {
  return launch_attn_fwd_nq<D, DROP, 1>(a, stream);
}

}  // namespace

static int attention_fwd_impl(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                              void* O, int ldo, float* lse2, const float* kbias, int B, int H, int Sq,
                              int Sk, int D, float scale, int causal, float p_drop, unsigned long long seed,
                              lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!Q || !K || !V || !O) return LC2IS_ERR_NULL;
  if (B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0) return LC2IS_ERR_SHAPE;
  if (ldq < H * D || ldk < H * D || ldv < H * D || ldo < H * D || ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4)
    return LC2IS_ERR_SHAPE;
  if (causal && Sq != Sk) return LC2IS_ERR_UNSUPPORTED;
  if (!(p_drop >= 0.f && p_drop < 1.f)) return LC2IS_ERR_UNSUPPORTED;
  if ((double)B * Sq * ldq * 2.0 >= 2147483648.0 || (double)B * (Sk + 64) * ldk * 2.0 >= 2147483648.0 ||
      (double)B * (Sk + 64) * ldv * 2.0 >= 2147483648.0)
    return LC2IS_ERR_UNSUPPORTED;
  AttnFwdArgs a{(const bf16_t*)Q, ldq, (const bf16_t*)K, ldk, (const bf16_t*)V, ldv, (bf16_t*)O, ldo, lse2,
                kbias, B, H, Sq, Sk, scale * 1.44269504088896341f, causal, make_drop_cfg(p_drop, seed)};
  if (a.drop.thr) {
    if ((double)B * H * Sq >= 4294967296.0) return LC2IS_ERR_UNSUPPORTED;   // 32-bit row coordinate of the RNG
    switch (D) {
      case 64: return launch_attn_fwd<64, true>(a, stream);
      case 96: return launch_attn_fwd<96, true>(a, stream);
      case 128: return launch_attn_fwd<128, true>(a, stream);
      default: return LC2IS_ERR_UNSUPPORTED;
    }
  }
  switch (D) {
    case 64: return launch_attn_fwd<64, false>(a, stream);
    case 96: return launch_attn_fwd<96, false>(a, stream);
    case 128: return launch_attn_fwd<128, false>(a, stream);
    default: return LC2IS_ERR_UNSUPPORTED;
  }
}

extern "C" int lc2is_attention_fwd(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                                   void* O, int ldo, float* lse2, const float* kbias, int B, int H, int Sq,
                                   int Sk, int D, float scale, int causal, lc2is_stream_t stream) {
  return attention_fwd_impl(Q, ldq, K, ldk, V, ldv, O, ldo, lse2, kbias, B, H, Sq, Sk, D, scale, causal, 0.f, 0ULL, stream);
}

extern "C" int lc2is_attention_fwd_dropout(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                                           void* O, int ldo, float* lse2, const float* kbias, int B, int H, int Sq,
                                           int Sk, int D, float scale, int causal, float p_drop,
                                           unsigned long long seed, lc2is_stream_t stream) {
  return attention_fwd_impl(Q, ldq, K, ldk, V, ldv, O, ldo, lse2, kbias, B, H, Sq, Sk, D, scale, causal, p_drop, seed,
                            stream);
}
