// Shared pieces of the attention kernels (attention_fwd.hip, attention_bwd.hip): the LDS image of a 64-row operand tile, its
// LDS-DMA fill, compile-time / run-time ring offsets and the LDS accessors that take 32-bit LDS addresses.
#pragma once
#include "common.h"
#include <type_traits>

namespace {

// K and V tiles (64 keys) live in LDS as ONE kind of image, filled by LDS-DMA (buffer_load ... lds, no staging registers, no
// ds_write) and read BOTH by rows (ds_read_b128: the K fragments of S^T) and transposed (ds_read_b64_tr_b16: the V^T fragments
// of O^T): rows of PITCH bytes, 16-byte chunk index XOR-swizzled by row bits so that both kinds of read are bank-conflict free
// (the image of attention_bwd.hip; rocprofv3 SQ_LDS_BANK_CONFLICT = 0 there).  An LDS-DMA wave instruction writes 1 KiB
// linearly, so the swizzle is applied on the per-lane SOURCE address and the reads use the same involution.
template <int D> struct AttnCfg {
  static constexpr int PITCH = (D == 64) ? 128 : 256;
  static constexpr int TILE = 64 * PITCH;               // one operand tile
  static constexpr int STAGE = 2 * TILE;                // K image | V image
  static constexpr int NSTAGE = (D == 64) ? 3 : 2;      // ring depth: tiles are requested NSTAGE-1 ahead (3 x 16 KiB / 2 x 32 KiB)
  static constexpr int CH = D / 8;                      // 16-byte chunks of data per row
  static constexpr int SLOTS = PITCH / 16;              // 16-byte slots per row
  static constexpr int RPP = 1024 / PITCH;              // rows per 1-KiB DMA piece
  static constexpr int PPW = 64 / RPP / 4;              // pieces per wave per operand tile
  static constexpr int IPT = 2 * PPW;                   // DMA instructions per wave per tile
  __device__ static __forceinline__ int swz(int row) {
    if constexpr (D == 64) return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
    else return ((row & 3) << 2) | ((row >> 2) & 3);
  }
  __device__ static __forceinline__ int off(int row, int ch) { return row * PITCH + ((ch ^ swz(row)) << 4); }
};

// one K / V (or Q / dO) tile pair: PPW 1-KiB pieces per operand and wave, requested by inline-asm LDS-DMA (common.h: lds_dma16)
template <int D>
__device__ __forceinline__ void attn_dma_tile(const bf16_t* K, unsigned kbytes, const bf16_t* V, unsigned vbytes, char* stage,
                                              int wid, const int* k_goff, const int* v_goff, int kstep, int vstep) {
  using Cfg = AttnCfg<D>;
  const __amdgpu_buffer_rsrc_t rsK = make_rsrc(K, kbytes);
  const __amdgpu_buffer_rsrc_t rsV = make_rsrc(V, vbytes);
  const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)LDS_PTR(stage));
  const int ks = __builtin_amdgcn_readfirstlane(kstep), vs = __builtin_amdgcn_readfirstlane(vstep);
#pragma unroll
  for (int j = 0; j < Cfg::PPW; ++j) {
    lds_dma16(rsK, base + (wid * Cfg::PPW + j) * 1024, k_goff[j], ks);
    lds_dma16(rsV, base + Cfg::TILE + (wid * Cfg::PPW + j) * 1024, v_goff[j], vs);
  }
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));
// a ring-stage byte offset that is either a compile-time constant (the unrolled main loop: it folds into the LDS instructions'
// offset fields and into m0) or a run-time value (the rolled loop of the leftover / masked tiles)
template <int V> using CtOff = std::integral_constant<int, V>;
struct RtOff {
  int v;
  __device__ constexpr operator int() const { return v; }
};

// LDS accesses by 32-bit LDS address (base register + immediate): the address of `extern __shared__` memory is only fixed
// when the module's static LDS is laid out, after instruction selection, so `smem + lane_offset + constant` costs a
// `v_add_u32 v, 0, v` per access; a lane's fragment bases are formed ONCE as opaque LDS addresses instead
__device__ __forceinline__ bf16x8_t lds_read_b128(unsigned addr) {
  return *(const __attribute__((address_space(3))) bf16x8_t*)(size_t)addr;
}
__device__ __forceinline__ bf16x8_t tr_frag2a(unsigned addr_lo, unsigned addr_hi) {
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(size_t)addr_lo);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(size_t)addr_hi);
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

__device__ __forceinline__ float max2f(float a, float b) {
  float d;
  asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float d;   // one VALU op for two comparisons (hipcc otherwise canonicalises the MFMA outputs before every fmaxf)
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}


}  // namespace
