// Fused attention backward in ONE main kernel with the algorithm's FIVE matrix products per tile
//   S = Q·K^T, dP = dO·V^T, dV^T += dO^T·P, dK^T += Q^T·dS, dQ^T += K^T·dS^T
// (attention_bwd.hip's two-launch form executes seven: S and dP once per launch, and streams Q / K / V / dO twice).
// replaces: autograd of hf eager_attention_forward (hf:modeling_clip.py:259-277, call site reference model/encoder.py:29-30)
//   and of torch multi_head_attention_forward's attention core (reference model/decoder.py:9-21), reached from
//   loss.backward() (reference engine.py:100).
//
// Work split.  A work ITEM is one block of KB = 128·NKT keys of one (batch, head) — a "member" of that (batch, head)'s CHAIN of
// nkb = ceil(Sk / KB) key blocks.  A workgroup (4 waves) holds dK^T / dV^T of its keys in accumulator registers (wave w: key
// tiles w·NKT .. w·NKT+NKT-1 of 32 keys, key on the MFMA lane) while it sweeps the 64-query tiles.  S and dP are computed with
// the key on the lane, so their fp32 accumulators, packed to bf16, ARE the B operands of dV^T and dK^T (accumulator as operand);
// only dS crosses LDS, once per tile: every wave writes its dS rows to a [key][query] image, and after one barrier wave w
// computes the 32x32 tile(s) (d-tile, query half) of dQ^T over ALL the block's keys (K^T fragments live in registers).
//
// dQ is summed over the members of a chain by an ORDERED HAND-OFF (bitwise reproducible, no float atomics: at D = 64 and 128-key
// blocks float atomics would need 2.3 TB/s of added bytes against a chip-wide atomic rate of 1.3): for query tile t the members
// add their tile to a running fp32 sum R[chain][t] in a fixed order — member x = t mod nkb first, then x+1, ... (mod nkb) — each
// wave for its own tile(s): poll the tile's per-wave flag (relaxed agent-scope loads) until its predecessor count is reached,
// one agent-scope acquire, write-through (sc1) loads of R, add, write-through stores of R (or, for the last member, scale and
// store the bf16 dQ rows), a counted s_waitcnt that covers the stores, then a relaxed agent-scope store of the flag.  The tile
// ORDER of a member is a Latin square over (member, time): with G = ceil(nqt / nkb) tile groups, member j at slot s works on tile
// (s mod G)·nkb + ((j - s / G) mod nkb), i.e. it is at position s / G of that tile's chain and its predecessor had the tile G
// slots earlier — every member starts at the head of some chain (no pipeline fill), and a hand-off has G - 1 slots of slack.
// Causal launches drop the members whose keys all lie above a tile's queries from that tile's chain.
//
// Items are drawn from per-XCD ticket queues (a chain's members run on one XCD — they stream the same Q / dO through that L2;
// placement is a speed matter only) by a PERSISTENT grid, exhausted queues are helped out by the other XCDs' workgroups.
// Progress does not depend on dispatch order or placement: tickets of a queue are taken in order, so at most one chain per
// queue is partially taken; with 8 (nkb - 1) < resident workgroups some workgroup is always free to take the missing member
// (the launcher refuses nkb > 32 and the two-launch form takes over).  Every spin is bounded; a timeout sets the state block's
// error word (checked by the tests and by ops.attention_bwd_status()) and lets the kernel drain.
// delta = rowsum(dO∘O) comes from a streaming prologue launch (lc2is_attention_delta).
#include "attn_common.h"
#include "lc2is_hip.h"
#include <cstdlib>

namespace {

struct AttnBwdFArgs {
  const bf16_t* Q; int ldq;
  const bf16_t* K; int ldk;
  const bf16_t* V; int ldv;
  const bf16_t* dO; int lddo;
  bf16_t* dQ; int lddq;
  bf16_t* dK; int lddk;
  bf16_t* dV; int lddv;
  const float* lse2;   // [B,H,Sq]
  const float* delta;  // [B,H,Sq]
  const float* kbias;  // [B,Sk] or null
  float* R;            // running dQ sums, fragment order: [chain][tile][2*NDT][4][64] x 16 bytes
  unsigned* state;     // [0..7] queue heads, [8] error word, [16 ..] flags [chain][tile][4 waves]
  int B, H, Sq, Sk;
  float scale, scale_log2;
  int causal;
  int nkb, nqt, G, nchains;
};

constexpr float LOG2E_F = 1.44269504088896341f;
constexpr int FUSED_STATE_HDR = 16;          // words in front of the flags
constexpr unsigned FUSED_SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ bf16x8_t pack8f(const f32x16_t& v, int base) {
  bf16x8_t r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)v[base + j];
  return r;
}

__device__ __forceinline__ bf16x8_t tr_frag_lds(unsigned addr_lo, unsigned addr_hi) {
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(size_t)addr_lo);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(size_t)addr_hi);
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// dS image [key][query]: 128-byte rows (64 queries), 8-byte chunk index XOR-swizzled by the key's low bits so that the
// ds_write_b64 of 16 consecutive keys and the transposed reads of 4 keys x 32 queries are both bank-conflict free
__device__ __forceinline__ int ds_swz(int key) { return (((key >> 1) & 1) << 3) | ((key & 1) << 2) | ((key >> 2) & 3); }

// LDS-DMA of the K block (KB rows) into the image of attn_common.h, rows past Sk zero-filled (their dS is 0, but 0 x garbage
// must stay 0); helper without a buffer-resource parameter (see attn_common.h)
template <int D, int KB>
__device__ __forceinline__ void dma_k_block(const bf16_t* K, unsigned kbytes, char* dst, int wid, int lane, int b, int head,
                                            int key0, int Sk, int ldk) {
  using Cfg = AttnCfg<D>;
  constexpr int KPW = KB / Cfg::RPP / 4;
  const __amdgpu_buffer_rsrc_t rsK = make_rsrc(K, kbytes);
#pragma unroll
  for (int j = 0; j < KPW; ++j) {
    const int piece = wid * KPW + j;
    const int row = Cfg::RPP * piece + lane / Cfg::SLOTS;
    const int ch = (lane % Cfg::SLOTS) ^ Cfg::swz(row);
    const int key = key0 + row;
    const int off = (ch < Cfg::CH && key < Sk) ? ((b * Sk + key) * ldk + head * D + ch * 8) * 2 : -1;
    lds_dma16(rsK, __builtin_amdgcn_readfirstlane((unsigned)(size_t)LDS_PTR(dst)) + piece * 1024, off, 0);
  }
}

template <int D, int NKT>
__global__ __launch_bounds__(256, (D == 64 && NKT == 1) ? 2 : 1) void attn_bwd_fused_kernel(AttnBwdFArgs p) {
  using I = AttnCfg<D>;
  constexpr int NKS = D / 16, NDT = D / 32;
  constexpr int KB = 128 * NKT;                    // keys per workgroup
  constexpr int KSTEPS = KB / 16;                  // 16-key steps of the dQ product
  constexpr int NT = 2 * NDT;                      // dQ^T tiles of a 64-query step: (d-tile, query half)
  constexpr int TPW = (NT + 3) / 4;                // ... per wave
  constexpr int STAGE = 2 * I::TILE + 768;         // Q image, dO image, 64 lse2, 64 delta (+ 64 spare words)
  constexpr int OFF_DS = 2 * STAGE;                // dS image [KB keys][64 queries]
  constexpr int OFF_K = OFF_DS + KB * 128;         // the block's K rows (image of attn_common.h), read transposed by the dQ product
  constexpr int OFF_MISC = OFF_K + KB * I::PITCH;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, l31 = lane & 31;
  const int q4 = (lane >> 2) & 3, p4 = lane & 3, cg = (lane >> 4) & 1;
  const float INF = __builtin_inff();
  const unsigned smem_a = (unsigned)(size_t)LDS_PTR(smem);
  const int xcc = (int)(__builtin_amdgcn_s_getreg(6164) & 7u);   // hwreg(HW_REG_XCC_ID, 0, 4): which queue to draw from first

  // lane-constant LDS addresses (offsets inside a stage / image)
  int row_addr[NKS], tr_lo[NDT], tr_hi[NDT];
#pragma unroll
  for (int s = 0; s < NKS; ++s) row_addr[s] = I::off(l31, 2 * s + hh);
#pragma unroll
  for (int d = 0; d < NDT; ++d) {
    const int e = 32 * d + 16 * cg + 4 * p4;
    tr_lo[d] = I::off(4 * hh + q4, e >> 3) + (e & 7) * 2;
    tr_hi[d] = I::off(4 * hh + q4 + 8, e >> 3) + (e & 7) * 2;
  }
  // dS image: write offsets of this wave's key rows, read addresses of its dQ tile(s); K^T read addresses of the same tile(s)
  unsigned ds_wr[NKT];   // byte offset of chunk hh of the key's row; chunk c0 + hh lives at smem_a + (ds_wr ^ (c0 << 3))
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    const int kl = 32 * (wid * NKT + kt) + l31;
    ds_wr[kt] = OFF_DS + kl * 128 + (((ds_swz(kl) ^ hh) & 15) << 3);
  }
  unsigned ds_rd_lo[TPW], ds_rd_hi[TPW], kt_lo[TPW], kt_hi[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int id = wid + 4 * i, qh = id & 1, dt = (id >> 1) < NDT ? (id >> 1) : 0;
    const int kl = 8 * hh + q4;                      // + 16 ks (both swizzles only look at key & 15)
    const int c = 8 * qh + 4 * cg + p4;
    ds_rd_lo[i] = smem_a + OFF_DS + kl * 128 + ((c ^ ds_swz(kl)) << 3);
    ds_rd_hi[i] = smem_a + OFF_DS + (kl + 4) * 128 + ((c ^ ds_swz(kl + 4)) << 3);
    const int e = 32 * dt + 16 * cg + 4 * p4;
    kt_lo[i] = smem_a + OFF_K + I::off(kl, e >> 3) + (e & 7) * 2;
    kt_hi[i] = smem_a + OFF_K + I::off(kl + 4, e >> 3) + (e & 7) * 2;
  }

  const unsigned qbytes = (unsigned)p.B * p.Sq * p.ldq * 2u, gbytes = (unsigned)p.B * p.Sq * p.lddo * 2u;
  const unsigned kbytes = (unsigned)p.B * p.Sk * p.ldk * 2u, vbytes = (unsigned)p.B * p.Sk * p.ldv * 2u;
  const unsigned rbytes = (unsigned)p.nchains * (unsigned)p.nqt * (unsigned)(NT * 4096);
  unsigned* const err_word = p.state + 8;
  int* const misc = (int*)(smem + OFF_MISC);

  for (;;) {
    // ---- take a ticket: own XCD's queue first, then the others' ----
    if (tid == 0) {
      int chain = -1, member = 0;
      for (int a = 0; a < 8 && chain < 0; ++a) {
        const int x = (xcc + a) & 7;
        if (x >= p.nchains) continue;
        const int nq = (p.nchains - x + 7) >> 3;   // chains of queue x: x, x+8, ...
        const unsigned n = __hip_atomic_fetch_add(p.state + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n < (unsigned)(nq * p.nkb)) {
          chain = x + 8 * (int)(n / (unsigned)p.nkb);
          member = (int)(n % (unsigned)p.nkb);
        }
      }
      misc[0] = chain;
      misc[1] = member;
    }
    __syncthreads();
    const int chain = __builtin_amdgcn_readfirstlane(misc[0]);
    const int member = __builtin_amdgcn_readfirstlane(misc[1]);
    if (chain < 0) break;
    const int b = chain / p.H, head = chain % p.H;
    const int key0 = member * KB;
    const bool wave_active = key0 + wid * NKT * 32 < p.Sk;   // wave-uniform

    // ---- the block's K rows -> LDS image (stays for the whole item) ----
    dma_k_block<D, KB>(p.K, kbytes, smem + OFF_K, wid, lane, b, head, key0, p.Sk, p.ldk);

    int kcol[NKT];
    bf16x8_t kf[NKT][NKS], vf[NKT][NKS];
    float bias[NKT];
    {
      const __amdgpu_buffer_rsrc_t rsK = make_rsrc(p.K, kbytes);
      const __amdgpu_buffer_rsrc_t rsV = make_rsrc(p.V, vbytes);
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        kcol[kt] = key0 + (wid * NKT + kt) * 32 + l31;
        const bool kok = kcol[kt] < p.Sk;
        const int tok = b * p.Sk + kcol[kt];
        const int ko = (tok * p.ldk + head * D + 8 * hh) * 2, vo = (tok * p.ldv + head * D + 8 * hh) * 2;
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
          kf[kt][s] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rsK, kok ? ko + s * 32 : -1, 0, 0));
          vf[kt][s] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rsV, kok ? vo + s * 32 : -1, 0, 0));
        }
        bias[kt] = kok ? (p.kbias ? p.kbias[(size_t)b * p.Sk + kcol[kt]] * LOG2E_F : 0.f) : -INF;
      }
    }
    // inactive waves (ragged last block) never write their dS rows: zero them once
    if (!wave_active) {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        char* row = smem + OFF_DS + (32 * (wid * NKT + kt) + l31) * 128 + hh * 64;
#pragma unroll
        for (int c = 0; c < 4; ++c) *(i32x4_t*)(row + 16 * c) = i32x4_t{0, 0, 0, 0};
      }
    }

    // ---- Q / dO tile staging (attention_bwd.hip's dK/dV scheme) ----
    int q_goff[I::PPW], g_goff[I::PPW];
#pragma unroll
    for (int j = 0; j < I::PPW; ++j) {
      const int row = I::RPP * (wid * I::PPW + j) + lane / I::SLOTS;
      const int ch = (lane % I::SLOTS) ^ I::swz(row);
      q_goff[j] = ch < I::CH ? ((b * p.Sq + row) * p.ldq + head * D + ch * 8) * 2 : -1;
      g_goff[j] = ch < I::CH ? ((b * p.Sq + row) * p.lddo + head * D + ch * 8) * 2 : -1;
    }
    float rstat = 0.f;
    const float inv_sl2 = 1.f / p.scale_log2;
    auto gload = [&](int qt, char* stage) __attribute__((always_inline)) {
      attn_dma_tile<D>(p.Q, qbytes, p.dO, gbytes, stage, wid, q_goff, g_goff, qt * 64 * p.ldq * 2, qt * 64 * p.lddo * 2);
      if (tid < 128) {  // threads 0..63: lse2, 64..127: delta — the RAW values (transformed in lstore: nothing here may wait for them)
        const int q = qt * 64 + (tid & 63);
        const size_t si = ((size_t)b * p.H + head) * p.Sq + q;
        const float* src = tid < 64 ? p.lse2 : p.delta;
        rstat = (q < p.Sq) ? src[si] : (tid < 64 ? -INF : 0.f);
      }
    };
    auto lstore = [&](char* stage) __attribute__((always_inline)) {
      if (tid < 128) {
        // initial accumulator of S: -lse2 / scale_log2, so that exp2(scale_log2 * (S + init)) = P (-inf: empty / padded row);
        // initial accumulator of dP: -delta
        const float v = tid < 64 ? (rstat == -INF ? -INF : -rstat * inv_sl2) : -rstat;
        *(float*)(stage + 2 * I::TILE + tid * 4) = v;
      }
      wait_vm0();
    };

    // ---- schedule: slot s -> tile (or -1) ----
    const int nslots = p.G * p.nkb;
    auto tile_of = [&](int s) -> int {
      const int g = s % p.G, pos = s / p.G;
      int x = member - pos;
      if (x < 0) x += p.nkb;
      const int t = g * p.nkb + x;
      if (t >= p.nqt) return -1;
      if (p.causal && t * 64 + 63 < key0) return -1;   // every key of the block lies above every query of the tile
      return t;
    };

    f32x16_t dkt[NKT][NDT], dvt[NKT][NDT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkt[kt][d][r] = 0.f; dvt[kt][d][r] = 0.f; }

    int pub_tile = -1, pub_cnt = 0;                      // a tile whose sum is stored but whose flag is not yet published

    auto flag_of = [&](int t) -> unsigned* { return p.state + FUSED_STATE_HDR + ((size_t)chain * p.nqt + t) * 4 + wid; };
    auto r_off = [&](int t, int id) -> int { return (int)((((unsigned)chain * p.nqt + t) * NT + id) * 4096u) + lane * 16; };

    // position of this member in tile t's chain: predecessors before it and whether it is the last
    auto chain_pos = [&](int t, int& before, int& last) {
      const int x = t % p.nkb;
      if (!p.causal) {
        before = member - x;
        if (before < 0) before += p.nkb;
        last = before == p.nkb - 1;
      } else {
        int L = (t * 64 + 63) / KB;                  // members 0..L take part
        if (L > p.nkb - 1) L = p.nkb - 1;
        if (member >= x) {
          const int hi = (member - 1 < L) ? member - 1 : L;
          before = hi >= x ? hi - x + 1 : 0;
        } else {
          before = (L >= x ? L - x + 1 : 0) + member;
        }
        last = before == L;
      }
    };
    auto publish = [&]() __attribute__((always_inline)) {
      if (pub_tile >= 0) {
        if (lane == 0) __hip_atomic_store(flag_of(pub_tile), (unsigned)pub_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pub_tile = -1;
      }
    };

    // ---- one tile of 64 queries ----
    // FAST: no key of the block is past Sk or biased and the launch is not causal (no per-score control flow)
    auto tile_body = [&](int t, int tn, int it, auto fast_c) __attribute__((always_inline)) {
      constexpr bool FAST = decltype(fast_c)::value;
      char* nxt = smem + ((it + 1) & 1) * STAGE;
      const unsigned cur_a = smem_a + (it & 1) * STAGE;
      const unsigned stat_a = cur_a + 2 * I::TILE + 16 * hh;   // S initial values; dP initial values at + 256

      int before, last;
      chain_pos(t, before, last);
      const bool need_sum = before > 0;
      // the flag of this tile's predecessors: asked for now, looked at after the two halves
      unsigned fv = 0;
      if (need_sum) fv = __hip_atomic_load(flag_of(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tn >= 0) gload(tn, nxt);
      const bool diag = p.causal && (key0 + KB - 1 > t * 64);   // some key of the block may exceed some query of the tile

      if (wave_active) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          __builtin_amdgcn_sched_barrier(0);   // a half tile is the scheduling region
          if (u == 1 && t * 64 + 32 >= p.Sq) {   // ragged last tile of at most 32 queries: the second half is empty — zero its dS
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
              for (int g2 = 0; g2 < 4; ++g2)
                *(__attribute__((address_space(3))) i32x2_t*)(size_t)(smem_a + (ds_wr[kt] ^ (unsigned)((8 + 2 * g2) << 3))) = i32x2_t{0, 0};
            continue;
          }
          // every LDS operand of the half is requested up front: Q / dO rows (A operands of S / dP) and the rows' initial values
          bf16x8_t qfr[NKS], gfr[NKS];
#pragma unroll
          for (int s = 0; s < NKS; ++s) {
            const unsigned a = cur_a + row_addr[s] + 32 * u * I::PITCH;
            qfr[s] = lds_read_b128(a);
            gfr[s] = lds_read_b128(a + I::TILE);
          }
          f32x16_t si, di;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const f32x4_t l4 = *(const __attribute__((address_space(3))) f32x4_t*)(size_t)(stat_a + 128 * u + 32 * c);
            const f32x4_t d4 = *(const __attribute__((address_space(3))) f32x4_t*)(size_t)(stat_a + 256 + 128 * u + 32 * c);
#pragma unroll
            for (int j = 0; j < 4; ++j) { si[4 * c + j] = l4[j]; di[4 * c + j] = d4[j]; }
          }
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt) {
            f32x16_t sa = si, dp = di;
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
              sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[s], kf[kt][s], sa, 0, 0, 0);
              dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gfr[s], vf[kt][s], dp, 0, 0, 0);
            }
            // P = exp2(scale_log2 * S'), dS = P * dP'   (S', dP' already carry -lse and -delta)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float s2 = sa[r] * p.scale_log2;
              if constexpr (!FAST) {
                s2 += bias[kt];
                if (diag) {
                  const int q = t * 64 + 32 * u + 8 * (r >> 2) + 4 * hh + (r & 3);
                  if (kcol[kt] > q) s2 = -INF;
                }
              }
              const float pr = __builtin_amdgcn_exp2f(s2);
              sa[r] = pr;
              dp[r] = pr * dp[r];
            }
#pragma unroll
            for (int s2i = 0; s2i < 2; ++s2i) {
              const bf16x8_t pf = pack8f(sa, 8 * s2i);
              const bf16x8_t dsf = pack8f(dp, 8 * s2i);
              // dS rows of this wave's keys: queries 32u + 16 s2i + 4hh + {0..3} and + 8
              const i32x4_t dsw = __builtin_bit_cast(i32x4_t, dsf);
              const int c0 = 8 * u + 4 * s2i;
              *(__attribute__((address_space(3))) i32x2_t*)(size_t)(smem_a + (ds_wr[kt] ^ (unsigned)(c0 << 3))) = i32x2_t{dsw[0], dsw[1]};
              *(__attribute__((address_space(3))) i32x2_t*)(size_t)(smem_a + (ds_wr[kt] ^ (unsigned)((c0 + 2) << 3))) = i32x2_t{dsw[2], dsw[3]};
              const int roff = (32 * u + 16 * s2i) * I::PITCH;
#pragma unroll
              for (int d = 0; d < NDT; ++d) {
                const unsigned lo = cur_a + tr_lo[d] + roff, hi = cur_a + tr_hi[d] + roff;
                const bf16x8_t gtf = tr_frag_lds(lo + I::TILE, hi + I::TILE);   // dO^T
                const bf16x8_t qtf = tr_frag_lds(lo, hi);                       // Q^T
                dvt[kt][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gtf, pf, dvt[kt][d], 0, 0, 0);
                dkt[kt][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf, dkt[kt][d], 0, 0, 0);
              }
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // the previous tile's sum was stored before this tile's flag load and DMA requests were issued: a counted wait that
      // leaves only the youngest I::IPT vector-memory operations (DMA pieces) in flight covers those stores — publish its flag
      if (pub_tile >= 0) {
        if (tn >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(I::IPT) : "memory");
        else wait_vm0();
        publish();
      }
      // the predecessors' running sum of this tile: requested now, added after the dQ product
      f32x16_t rr[TPW];
      if (need_sum) {
        unsigned* f = flag_of(t);
        unsigned v = __builtin_amdgcn_readfirstlane(fv);
        unsigned spins = 0;
        while (v < (unsigned)before) {   // (nothing of this wave's is unpublished here: it cannot be part of a cycle)
          __builtin_amdgcn_s_sleep(2);
          v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          if ((++spins & 1023u) == 0) {
            const unsigned e = __builtin_amdgcn_readfirstlane(__hip_atomic_load(err_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (e != 0 || spins >= FUSED_SPIN_LIMIT) {
              if (lane == 0) __hip_atomic_store(err_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              break;
            }
          }
        }
        // every load of the handed-off bytes is a write-through-coherent (sc1) load: no L1 line of them can exist, so the
        // agent-scope acquire (an L1 invalidate per wave and tile: measured 3.6x the whole kernel) is not needed; the fence
        // below only keeps the compiler from moving the loads above the poll
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      {
        // (a wave without predecessors reads offset -1: the range check returns zeros — no branch around the loads, so the
        // compiler's wait for them sits at their first use, after the dQ product)
        const __amdgpu_buffer_rsrc_t rsR = make_rsrc(p.R, rbytes);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          const int id = wid + 4 * i;
          const int ro = (need_sum && id < NT) ? r_off(t, id) : -1;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const f32x4_t v4 = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsR, ro < 0 ? -1 : ro + c * 1024, 0, 16));
#pragma unroll
            for (int j = 0; j < 4; ++j) rr[i][4 * c + j] = v4[j];
          }
        }
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's dS writes are done
      __builtin_amdgcn_s_barrier();          // ... and so are every wave's
      __builtin_amdgcn_sched_barrier(0);
      // dQ^T tile(s) over all the block's keys
      f32x16_t dq[TPW];
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const int id = wid + 4 * i;
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;
        if (id < NT) {
#pragma unroll
          for (int ks = 0; ks < KSTEPS; ++ks) {
            const bf16x8_t ka = tr_frag_lds(kt_lo[i] + ks * 16 * I::PITCH, kt_hi[i] + ks * 16 * I::PITCH);
            const bf16x8_t dsb = tr_frag_lds(ds_rd_lo[i] + ks * 2048, ds_rd_hi[i] + ks * 2048);
            dq[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, dsb, dq[i], 0, 0, 0);
          }
        }
      }
      if (tn >= 0) lstore(nxt);
      else wait_vm0();
#pragma unroll
      for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i][r] += rr[i][r];
      __syncthreads();   // next stage landed; the dS image and this stage are free again
      // pass the sum on (write-through stores, published half a tile later) or, as the chain's last member, write dQ
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const int id = wid + 4 * i;
        if (id < NT) {
          if (last) {
            const __amdgpu_buffer_rsrc_t rsD = make_rsrc(p.dQ, (unsigned)p.B * p.Sq * p.lddq * 2u);
            const int dt = id >> 1, qh = id & 1;
            const int q = t * 64 + 32 * qh + l31;
            const int base = (q < p.Sq) ? ((b * p.Sq + q) * p.lddq + head * D + 32 * dt + 4 * hh) * 2 : -1;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const unsigned lo = pack_bf16x2(dq[i][4 * c] * p.scale, dq[i][4 * c + 1] * p.scale);
              const unsigned hi = pack_bf16x2(dq[i][4 * c + 2] * p.scale, dq[i][4 * c + 3] * p.scale);
              __builtin_amdgcn_raw_buffer_store_b64(i32x2_t{(int)lo, (int)hi}, rsD, base < 0 ? -1 : base + 16 * c, 0, 0);
            }
          } else {
            const __amdgpu_buffer_rsrc_t rsR = make_rsrc(p.R, rbytes);
            const int ro = r_off(t, id);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const f32x4_t v4 = {dq[i][4 * c], dq[i][4 * c + 1], dq[i][4 * c + 2], dq[i][4 * c + 3]};
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v4), rsR, ro + c * 1024, 0, 16);   // sc1: write-through
            }
          }
        }
      }
      if (!last) { pub_tile = t; pub_cnt = before + 1; }
    };

    int s = 0, t = -1;
    for (; s < nslots; ++s) {
      t = tile_of(s);
      if (t >= 0) break;
    }
    if (t >= 0) gload(t, smem);
    lstore(smem);      // (also: the K image has landed)
    __syncthreads();
    // block-uniform choice of the body: no key of the block past Sk or biased, no causal edge anywhere
    const bool fast_block = p.kbias == nullptr && key0 + KB <= p.Sk && !p.causal;
    int it = 0;
    auto sweep = [&](auto fast_c) __attribute__((always_inline)) {
      while (t >= 0) {
        int sn = s + 1, tn = -1;
        for (; sn < nslots; ++sn) {
          tn = tile_of(sn);
          if (tn >= 0) break;
        }
        tile_body(t, tn, it, fast_c);
        s = sn;
        t = tn;
        ++it;
      }
    };
    if (fast_block) sweep(std::true_type{});
    else sweep(std::false_type{});
    if (pub_tile >= 0) {   // the last tile's hand-off
      wait_vm0();
      publish();
    }

    // ---- dK, dV of this block ----
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kcol[kt] < p.Sk) {
        bf16_t* krow = p.dK + (size_t)(b * p.Sk + kcol[kt]) * p.lddk + head * D;
        bf16_t* vrow = p.dV + (size_t)(b * p.Sk + kcol[kt]) * p.lddv + head * D;
#pragma unroll
        for (int d = 0; d < NDT; ++d)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int col = 32 * d + 8 * c + 4 * hh;
            uint2 pk = make_uint2(pack_bf16x2(dkt[kt][d][4 * c] * p.scale, dkt[kt][d][4 * c + 1] * p.scale),
                                  pack_bf16x2(dkt[kt][d][4 * c + 2] * p.scale, dkt[kt][d][4 * c + 3] * p.scale));
            *reinterpret_cast<uint2*>(krow + col) = pk;
            uint2 pv = make_uint2(pack_bf16x2(dvt[kt][d][4 * c], dvt[kt][d][4 * c + 1]),
                                  pack_bf16x2(dvt[kt][d][4 * c + 2], dvt[kt][d][4 * c + 3]));
            *reinterpret_cast<uint2*>(vrow + col) = pv;
          }
      }
    }
    __syncthreads();   // misc / LDS are reused by the next item
  }
}

// delta[b,h,q] = sum_d O[b,q,h,d] * dO[b,q,h,d]: one streaming pass (16-byte loads, a thread per 8 elements, the D/8 partial
// sums of a (row, head) meet in LDS)
template <int D>
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ O, int ldo, const bf16_t* __restrict__ dO, int lddo,
                                                          float* __restrict__ delta, int B, int H, int Sq) {
  constexpr int CPH = D / 8;                 // chunks per head
  constexpr int HPB = 256 / CPH;             // (row, head) pairs per block pass
  __shared__ float part[256];
  const long npairs = (long)B * Sq * H;
  const int sub = threadIdx.x / CPH, ch = threadIdx.x % CPH;
  for (long base = (long)blockIdx.x * HPB; base < npairs; base += (long)gridDim.x * HPB) {
    const long pair = base + sub;
    float s = 0.f;
    if (sub < HPB && pair < npairs) {
      const long row = pair / H;
      const int h = (int)(pair % H);
      const i32x4_t a = *(const i32x4_t*)(O + row * ldo + h * D + ch * 8);
      const i32x4_t g = *(const i32x4_t*)(dO + row * lddo + h * D + ch * 8);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned x = (unsigned)a[i], y = (unsigned)g[i];
        s += bf16_to_f32((bf16_t)(x & 0xffff)) * bf16_to_f32((bf16_t)(y & 0xffff));
        s += bf16_to_f32((bf16_t)(x >> 16)) * bf16_to_f32((bf16_t)(y >> 16));
      }
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < HPB && base + threadIdx.x < npairs) {
      float tsum = 0.f;
#pragma unroll
      for (int i = 0; i < CPH; ++i) tsum += part[threadIdx.x * CPH + i];
      const long pr = base + threadIdx.x;
      const long row = pr / H;
      const int h = (int)(pr % H);
      const long bb = row / Sq, q = row % Sq;
      delta[(bb * H + h) * Sq + q] = tsum;
    }
    __syncthreads();
  }
}

struct FusedPlan {
  int kb, nkb, nqt, G, nchains, nt;
  size_t state_bytes, r_bytes;
};

inline FusedPlan fused_plan(int B, int H, int Sq, int Sk, int D, int nkt) {
  FusedPlan f;
  f.kb = 128 * nkt;
  f.nkb = (Sk + f.kb - 1) / f.kb;
  f.nqt = (Sq + 63) / 64;
  f.G = (f.nqt + f.nkb - 1) / f.nkb;
  f.nchains = B * H;
  f.nt = 2 * (D / 32);
  const size_t words = FUSED_STATE_HDR + (size_t)f.nchains * f.nqt * 4;
  f.state_bytes = (words * 4 + 255) & ~(size_t)255;
  f.r_bytes = (size_t)f.nchains * f.nqt * f.nt * 4096;
  return f;
}

int fused_nkt() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("LC2IS_ATTN_BWD_NKT");
    v = (e && atoi(e) == 2) ? 2 : 1;
  }
  return v;
}

template <int D, int NKT>
int launch_fused(const AttnBwdFArgs& a, hipStream_t stream) {
  using I = AttnCfg<D>;
  constexpr int STAGE = 2 * I::TILE + 768;
  constexpr int LDS = 2 * STAGE + 128 * NKT * 128 + 128 * NKT * I::PITCH + 16;
  auto kern = attn_bwd_fused_kernel<D, NKT>;
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return LC2IS_ERR_LAUNCH;
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
  }
  const int per_cu = (D == 64 && NKT == 1) ? 2 : 1;
  long items = (long)a.nchains * a.nkb;
  long grid = (long)ncu * per_cu;
  if (grid > items) grid = items;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), LDS, stream, a);
  return lc2is_check_launch();
}

}  // namespace

extern "C" int lc2is_attention_delta(const void* O, int ldo, const void* dO, int lddo, float* delta, int B, int H, int Sq, int D,
                                     lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!O || !dO || !delta) return LC2IS_ERR_NULL;
  if (B <= 0 || H <= 0 || Sq <= 0) return LC2IS_ERR_SHAPE;
  if (ldo < H * D || lddo < H * D || ldo % 8 || lddo % 8) return LC2IS_ERR_SHAPE;
  const long npairs = (long)B * Sq * H;
  const int hpb = 256 / (D / 8);
  long blocks = (npairs + hpb - 1) / hpb;
  if (blocks > 4096) blocks = 4096;
  switch (D) {
    case 64: hipLaunchKernelGGL(attn_delta_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16_t*)O, ldo, (const bf16_t*)dO, lddo, delta, B, H, Sq); break;
    case 96: hipLaunchKernelGGL(attn_delta_kernel<96>, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16_t*)O, ldo, (const bf16_t*)dO, lddo, delta, B, H, Sq); break;
    case 128: hipLaunchKernelGGL(attn_delta_kernel<128>, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16_t*)O, ldo, (const bf16_t*)dO, lddo, delta, B, H, Sq); break;
    default: return LC2IS_ERR_UNSUPPORTED;
  }
  return lc2is_check_launch();
}

extern "C" size_t lc2is_attention_bwd_fused_workspace_bytes(int B, int H, int Sq, int Sk, int D) {
  if (B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0 || (D != 64 && D != 96 && D != 128)) return 0;
  const FusedPlan f = fused_plan(B, H, Sq, Sk, D, D == 64 ? fused_nkt() : 1);
  if (f.nkb > 32) return 0;                                   // progress bound of the ticket queues (file header)
  if ((double)f.r_bytes >= 2147483648.0) return 0;            // 32-bit buffer offsets
  return f.state_bytes + f.r_bytes;
}

extern "C" int lc2is_attention_bwd_fused(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                                         const void* O, int ldo, const void* dO, int lddo, void* dQ, int lddq, void* dK,
                                         int lddk, void* dV, int lddv, const float* lse2, float* delta, const float* kbias,
                                         int B, int H, int Sq, int Sk, int D, float scale, int causal, void* workspace,
                                         size_t workspace_bytes, lc2is_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!Q || !K || !V || !O || !dO || !dQ || !dK || !dV || !lse2 || !delta || !workspace) return LC2IS_ERR_NULL;
  if (B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0) return LC2IS_ERR_SHAPE;
  if (D != 64 && D != 96 && D != 128) return LC2IS_ERR_UNSUPPORTED;
  const int hd = H * D;
  if (ldq < hd || ldk < hd || ldv < hd || ldo < hd || lddo < hd || lddq < hd || lddk < hd || lddv < hd) return LC2IS_ERR_SHAPE;
  if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 8 || lddo % 8 || lddq % 4 || lddk % 4 || lddv % 4) return LC2IS_ERR_SHAPE;
  if (causal && Sq != Sk) return LC2IS_ERR_UNSUPPORTED;
  const double lim = 2147483648.0;
  if ((double)B * (Sq + 64) * ldq * 2.0 >= lim || (double)B * (Sq + 64) * lddo * 2.0 >= lim ||
      (double)B * (Sq + 64) * ldo * 2.0 >= lim || (double)B * (Sk + 256) * ldk * 2.0 >= lim ||
      (double)B * (Sk + 256) * ldv * 2.0 >= lim || (double)B * (Sq + 64) * lddq * 2.0 >= lim)
    return LC2IS_ERR_UNSUPPORTED;
  const int nkt = D == 64 ? fused_nkt() : 1;
  const FusedPlan f = fused_plan(B, H, Sq, Sk, D, nkt);
  if (f.nkb > 32 || (double)f.r_bytes >= lim) return LC2IS_ERR_UNSUPPORTED;
  if (workspace_bytes < f.state_bytes + f.r_bytes) return LC2IS_ERR_WORKSPACE;
  if (((size_t)workspace & 255) != 0) return LC2IS_ERR_SHAPE;

  int rc = lc2is_attention_delta(O, ldo, dO, lddo, delta, B, H, Sq, D, stream_);
  if (rc) return rc;
  // queue heads, error word and flags start every call at zero (a memset node under graph capture)
  if (hipMemsetAsync(workspace, 0, f.state_bytes, stream) != hipSuccess) return LC2IS_ERR_LAUNCH;
  AttnBwdFArgs a{(const bf16_t*)Q, ldq, (const bf16_t*)K, ldk, (const bf16_t*)V, ldv, (const bf16_t*)dO, lddo,
                 (bf16_t*)dQ, lddq, (bf16_t*)dK, lddk, (bf16_t*)dV, lddv, lse2, delta, kbias,
                 (float*)((char*)workspace + f.state_bytes), (unsigned*)workspace, B, H, Sq, Sk, scale, scale * LOG2E_F,
                 causal, f.nkb, f.nqt, f.G, f.nchains};
  switch (D) {
    case 64: return nkt == 2 ? launch_fused<64, 2>(a, stream) : launch_fused<64, 1>(a, stream);
    case 96: return launch_fused<96, 1>(a, stream);
    default: return launch_fused<128, 1>(a, stream);
  }
}

// error word of the last fused launch on this workspace (0 = every hand-off completed); the caller synchronises first
extern "C" int lc2is_attention_bwd_fused_status(const void* workspace, lc2is_stream_t stream_) {
  unsigned v = 0;
  if (!workspace) return LC2IS_ERR_NULL;
  if (hipMemcpyAsync(&v, (const char*)workspace + 32, 4, hipMemcpyDeviceToHost, (hipStream_t)stream_) != hipSuccess) return LC2IS_ERR_LAUNCH;
  if (hipStreamSynchronize((hipStream_t)stream_) != hipSuccess) return LC2IS_ERR_LAUNCH;
  return v == 0 ? LC2IS_OK : LC2IS_ERR_LAUNCH;
}
