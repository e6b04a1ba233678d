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
// query per 32-key sub-tile: the online-softmax max/sum are per-lane loops plus one lane^32 exchange, and the
// fp32 score registers, packed pairwise to bf16, are already the B operand of O^T[d][query] += V^T·P^T
// (accumulator-as-operand, no LDS round trip).  V^T fragments come from the row-major V tile with
// ds_read_b64_tr_b16.  K/V tiles of 64 keys are staged global->VGPR->LDS (range-checked loads), double
// buffered, K rows padded by 16 B (conflict-free ds_read_b128), V rows at a 192/320-byte pitch
// (conflict-free transposed reads).  Masks (key tail, key padding, causal) are an additive bias staged per
// tile and only applied on tiles that need one.
#include "common.h"
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

template <int D> struct AttnCfg {
  static constexpr int KS = D * 2 + 16;                 // K row pitch (bytes)
  static constexpr int VS = (D <= 96) ? 192 : 320;      // V row pitch: == 64 or 192 (mod 256), >= 2D
  static constexpr int KT = 64 * KS, VT = 64 * VS;
  static constexpr int STAGE = KT + VT + 256;           // + 64 fp32 bias values
  static constexpr int CH = D / 8;                      // 16-byte chunks per row
  static constexpr int NCH = 64 * CH / 256;             // chunks per thread per operand tile
};

__device__ __forceinline__ bf16x8_t tr_frag2(const char* base, int addr_lo, int addr_hi) {
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4_t*)LDS_PTR(base + addr_lo));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4_t*)LDS_PTR(base + addr_hi));
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// DBG (diagnostic builds, LC2IS_ATTN_DBG; results are WRONG by design): bit 0 = no softmax arithmetic, bit 1 = K / V^T fragments
// read from LDS once before the loop, bit 2 = no global loads / LDS stores inside the loop, bit 3 = no P.V MFMAs
// DROP: dropout on the attention probabilities (torch multi_head_attention_forward dropout_p in training mode,
// torch:nn/functional.py:6206): the normaliser l sums the UNDROPPED probabilities, O accumulates keep * P / (1 - p);
// coordinate of a score = (row (b*H + h)*Sq + q, column key), so the backward kernels regenerate the same decisions.
template <int D, int DBG = 0, bool DROP = false>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnFwdArgs p) {
  using Cfg = AttnCfg<D>;
  constexpr int KS = Cfg::KS, VS = Cfg::VS, CH = Cfg::CH, NCH = Cfg::NCH;
  constexpr int NKS = D / 16, NDT = D / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int hh = lane >> 5, l31 = lane & 31;
  // 1-D grid, XCD-aware: the query blocks of one (batch, head) — which stream the same K / V — are neighbours in the tile
  // order, and xcd_remap gives every XCD (its own L2) one contiguous chunk of that order
  const int nqb = (p.Sq + 127) / 128;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int bx = tile % nqb, head = (tile / nqb) % p.H, b = tile / (nqb * p.H);
  const int q0 = bx * 128 + wid * 32;
  const int qrow = q0 + l31;
  const bool wave_active = q0 < p.Sq;   // wave-uniform (wid comes from threadIdx.x >> 6)
  const float NEG_INF = -__builtin_inff();
  const unsigned drop_rh = DROP ? drop_row_hash(p.drop, (unsigned)((b * p.H + head) * p.Sq + qrow)) : 0u;

  int nkt = (p.Sk + 63) / 64;
  if (p.causal) {
    const int lim = (bx * 128 + 128 + 63) / 64;  // keys <= last query of the block
    if (lim < nkt) nkt = lim;
  }

  const unsigned qbytes = (unsigned)p.B * p.Sq * p.ldq * 2u;
  const unsigned kbytes = (unsigned)p.B * p.Sk * p.ldk * 2u, vbytes = (unsigned)p.B * p.Sk * p.ldv * 2u;
  const __amdgpu_buffer_rsrc_t rsQ = make_rsrc(p.Q, qbytes);
  const __amdgpu_buffer_rsrc_t rsK = make_rsrc(p.K, kbytes);
  const __amdgpu_buffer_rsrc_t rsV = make_rsrc(p.V, vbytes);

  // Q fragments (B operand of S^T = K·Q^T): lane holds Q[qrow][16s + 8hh .. +7]
  bf16x8_t qf[NKS];
  {
    const int off = (qrow < p.Sq) ? ((b * p.Sq + qrow) * p.ldq + head * D + 8 * hh) * 2 : -1;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      const i32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsQ, off < 0 ? -1 : off + s * 32, 0, 0);
      qf[s] = __builtin_bit_cast(bf16x8_t, v);
    }
  }

  // staging bookkeeping
  int k_goff[NCH], v_goff[NCH], k_lds[NCH], v_lds[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * 256, row = c / CH, ch = c % CH;
    k_goff[i] = ((b * p.Sk + row) * p.ldk + head * D + ch * 8) * 2;
    v_goff[i] = ((b * p.Sk + row) * p.ldv + head * D + ch * 8) * 2;
    k_lds[i] = row * KS + ch * 16;
    v_lds[i] = Cfg::KT + row * VS + ch * 16;
  }
  i32x4_t rk[NCH], rv[NCH];
  float rbias = 0.f;
  auto gload = [&](int kt) {
    const int kb = kt * 64 * p.ldk * 2, vb = kt * 64 * p.ldv * 2;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      rk[i] = __builtin_amdgcn_raw_buffer_load_b128(rsK, k_goff[i] + kb, 0, 0);
      rv[i] = __builtin_amdgcn_raw_buffer_load_b128(rsV, v_goff[i] + vb, 0, 0);
    }
    if (tid < 64) {
      const int key = kt * 64 + tid;
      rbias = (key < p.Sk) ? (p.kbias ? p.kbias[(size_t)b * p.Sk + key] * 1.44269504088896341f : 0.f) : NEG_INF;
    }
  };
  auto lstore = [&](char* stage) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      *(i32x4_t*)(stage + k_lds[i]) = rk[i];
      *(i32x4_t*)(stage + v_lds[i]) = rv[i];
    }
    if (tid < 64) *(float*)(stage + Cfg::KT + Cfg::VT + tid * 4) = rbias;
  };

  // fragment addresses
  const int k_frag = l31 * KS + 16 * hh;  // + 32t*KS + 32*s
  const int q4 = (lane >> 2) & 3, p4 = lane & 3, cg = (lane >> 4) & 1;
  const int v_frag = (4 * hh + q4) * VS + (16 * cg + 4 * p4) * 2;  // + (32t+16s2)*VS + 64*dt ; hi = +8*VS

  f32x16_t ot[NDT];
#pragma unroll
  for (int d = 0; d < NDT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[d][r] = 0.f;
  float m_run = NEG_INF, l_run = 0.f;

  if (nkt > 0) {
    gload(0);
    lstore(smem);
  }
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const char* cur = smem + (kt & 1) * Cfg::STAGE;
    char* nxt = smem + ((kt + 1) & 1) * Cfg::STAGE;
    const bool more = (kt + 1) < nkt;
    if (more && !(DBG & 4)) gload(kt + 1);

    // A wave whose 32 query rows all lie past Sq (the ragged last block of S = 1025: three of its four waves) only helps
    // staging the K/V tiles: one wave-uniform branch around the whole compute segment (branches INSIDE it hurt scheduling).
    if (wave_active) {
    // ---- S^T = K · Q^T (two 32-key sub-tiles) ----
    f32x16_t st[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        bf16x8_t kf = *(const bf16x8_t*)(((DBG & 2) ? smem : cur) + k_frag + 32 * t * KS + 32 * s);
        if (DBG & 2) { kf = qf[s]; asm volatile("" : "+v"(kf)); }
        st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[t], 0, 0, 0);
      }
    }

    // ---- softmax update: interior tiles (no bias, no causal edge, full 64 keys) take a branch-free path with
    // one fma + one exp per score; edge tiles take the general masked path ----
    const bool tail = (kt * 64 + 64 > p.Sk);
    const bool diag = p.causal && (kt * 64 + 63 > bx * 128);  // some key may exceed some query
    const bool masked = tail || diag || (p.kbias != nullptr);
    float alpha, psum = 0.f;
    if (DBG & 1) {
      alpha = 1.f;
    } else if (!masked) {
      float mx = st[0][0];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t][r]);
      mx *= p.scale_log2;  // scale > 0
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      const float neg_m = -m_new;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[t][r], p.scale_log2, neg_m));
          st[t][r] = e;
          psum += e;
        }
    } else {
      const float* biasv = (const float*)(cur + Cfg::KT + Cfg::VT);
      float mx = NEG_INF;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4_t bz = *(const f32x4_t*)(biasv + 32 * t + 8 * c + 4 * hh);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float sc = st[t][4 * c + j] * p.scale_log2 + bz[j];
            if (diag) {
              const int key = kt * 64 + 32 * t + 8 * c + 4 * hh + j;
              if (key > qrow) sc = NEG_INF;
            }
            st[t][4 * c + j] = sc;
            mx = fmaxf(mx, sc);
          }
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float m_use = (m_new == NEG_INF) ? 0.f : m_new;
      alpha = __builtin_amdgcn_exp2f(m_run - m_use);
      m_run = m_new;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __builtin_amdgcn_exp2f(st[t][r] - m_use);
          st[t][r] = e;
          psum += e;
        }
    }
    l_run = l_run * alpha + psum;
    if constexpr (DROP) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned key = (unsigned)(kt * 64 + 32 * t + 8 * (r >> 2) + 4 * hh + (r & 3));
          st[t][r] = drop_keep(p.drop, drop_rh, key) ? st[t][r] * p.drop.inv_keep : 0.f;
        }
    }
    if (__any(alpha != 1.f)) {  // lazy rescale: once the running max has settled nothing is multiplied
#pragma unroll
      for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[d][r] *= alpha;
    }

    // ---- O^T += V^T · P^T ----
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int t = ks >> 1, s2 = ks & 1;
      bf16x8_t pf;
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = (__bf16)st[t][8 * s2 + j];
#pragma unroll
      for (int d = 0; d < NDT; ++d) {
        const int a = v_frag + (32 * t + 16 * s2) * VS + 64 * d;
        bf16x8_t vf;
        if (DBG & 2) { vf = qf[d]; asm volatile("" : "+v"(vf)); }
        else vf = tr_frag2(cur + Cfg::KT, a, a + 8 * VS);
        if (!(DBG & 8)) ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, ot[d], 0, 0, 0);
        else ot[d][0] += (float)pf[0] + (float)vf[0];
      }
    }
    }  // wave_active

    if (more && !(DBG & 4)) lstore(nxt);
    __syncthreads();
  }

  // ---- normalise and store: lane = query, registers = 4-wide runs of d ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
  if (qrow < p.Sq) {
    bf16_t* orow = p.O + (size_t)(b * p.Sq + qrow) * p.ldo + head * D;
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int col = 32 * d + 8 * c + 4 * hh;
        uint2 pk = make_uint2(pack_bf16x2(ot[d][4 * c] * inv, ot[d][4 * c + 1] * inv),
                              pack_bf16x2(ot[d][4 * c + 2] * inv, ot[d][4 * c + 3] * inv));
        *reinterpret_cast<uint2*>(orow + col) = pk;
      }
    if (p.lse2 && hh == 0)
      p.lse2[((size_t)b * p.H + head) * p.Sq + qrow] =
          l_tot > 0.f ? m_run + __builtin_amdgcn_logf(l_tot) : NEG_INF;
  }
}

template <int D, int DBG = 0, bool DROP = false>
int launch_attn_fwd(const AttnFwdArgs& a, hipStream_t stream) {
  using Cfg = AttnCfg<D>;
  auto kern = attn_fwd_kernel<D, DBG, DROP>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * Cfg::STAGE) !=
        hipSuccess)
      return LC2IS_ERR_LAUNCH;
    attr_set = true;
  }
  static const int lds_pad = getenv("LC2IS_ATTN_LDS_PAD") ? atoi(getenv("LC2IS_ATTN_LDS_PAD")) : 0;   // occupancy probe (tools/attn_ablate.py)
  if (lds_pad) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * Cfg::STAGE + lds_pad);
  hipLaunchKernelGGL(kern, dim3(((a.Sq + 127) / 128) * a.H * a.B), dim3(256), 2 * Cfg::STAGE + lds_pad, stream, a);
  return lc2is_check_launch();
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
      case 64: return launch_attn_fwd<64, 0, true>(a, stream);
      case 96: return launch_attn_fwd<96, 0, true>(a, stream);
      case 128: return launch_attn_fwd<128, 0, true>(a, stream);
      default: return LC2IS_ERR_UNSUPPORTED;
    }
  }
  switch (D) {
    case 64: {
      static const int dbg = getenv("LC2IS_ATTN_DBG") ? atoi(getenv("LC2IS_ATTN_DBG")) : 0;
      switch (dbg) {
        case 1: return launch_attn_fwd<64, 1>(a, stream);
        case 2: return launch_attn_fwd<64, 2>(a, stream);
        case 4: return launch_attn_fwd<64, 4>(a, stream);
        case 6: return launch_attn_fwd<64, 6>(a, stream);
        case 7: return launch_attn_fwd<64, 7>(a, stream);
        case 8: return launch_attn_fwd<64, 8>(a, stream);
        case 15: return launch_attn_fwd<64, 15>(a, stream);
        default: return launch_attn_fwd<64>(a, stream);
      }
    }
    case 96: return launch_attn_fwd<96>(a, stream);
    case 128: return launch_attn_fwd<128>(a, stream);
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
