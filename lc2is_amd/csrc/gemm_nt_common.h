// Shared pieces of the NT GEMM kernels (gemm_nt.hip; the experimental kernels under tools/probes/gemm_nt_p192/ include it too):
// the argument block and the activation functions of the epilogues.
#pragma once
#include "common.h"
#include "lc2is_hip.h"

namespace {

struct GemmNtArgs {
  const bf16_t* A; int lda;
  const bf16_t* W; int ldw;
  const float* bias;
  const float* resid; int ldr;
  const bf16_t* aux_in; int ldx;
  bf16_t* out_bf16; int ldo;
  float* out_f32; int ldf;
  bf16_t* aux_out; int ldy;
  int M, N, K, act;
  int staged_epi;  // 1: bf16 epilogue traffic through LDS (needs N % 8 == 0 and 8-element-aligned leading dims);
                   // 2 (host side only, cleared before the launch): fp32-only output through LDS (F32EPI instantiations)
  // strided-batched form (gemm_nt_kernel only; blockIdx.y = batch): element strides between consecutive problems
  long bsA, bsW, bsOb, bsOf;
  // ragged rows folded into a large-tile launch (256x384 kernels): rows M .. M + tail_rows - 1 of the same A / output buffers are
  // computed by the blocks' waves after their own tile, one 16 x 16 fragment each (the rows kernel's algorithm, no second launch)
  int tail_rows;
  // LayerNorm of the fp32 output row fused into the 256x384 kernel's epilogue (EPI = -3, round 5): see gemm_nt_w384_kernel
  const float* ln_gamma; const float* ln_beta;
  bf16_t* ln_out; int ldl;
  float* ln_mean; float* ln_rstd;
  unsigned long long* ln_xchg;   // M x (N / 384) x 2 eight-byte {value, tag} granules, all zero between launches
  float ln_eps;
};

__device__ __forceinline__ float sigmoidf_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }  // v_exp + v_rcp
// sigmoid(1.702 x) of quick_gelu (hf:activations.py QuickGELUActivation) with the two constants of exp(-1.702 x) = exp2(-1.702 log2(e) x)
// folded into ONE multiply (round 5: sigmoidf_fast(1.702f * x) cost two — hipcc may not reassociate float products — in an epilogue
// whose cost IS its vector arithmetic: 44 -> 40 issue cycles per element beside the 128 accumulators of a lane)
__device__ __forceinline__ float quick_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -2.4554669596f)); }
// exact GELU of hf:activations.py "gelu" (Swin MLP, modeling_swin.py:474): 0.5 x (1 + erf(x / sqrt 2)) and its derivative
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
// derivative = Phi(x) + x phi(x).  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 gradient it scales):
// its exp(-(x/sqrt2)^2) IS the exp(-x^2/2) of phi, so the whole derivative is one v_exp, one v_rcp and seven FMAs — libm's
// branchy erff() here made the DGELU_ERF epilogue spill 418 VGPRs (231 us per launch against 64 for the forward one).
__device__ __forceinline__ float dgelu_erf(float x) {
  const float e = __expf(-0.5f * x * x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * 0.70710678118654752f * __builtin_fabsf(x));
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * e;
  return 0.5f * (1.0f + __builtin_copysignf(erf_abs, x)) + x * 0.39894228040143268f * e;
}

}  // namespace

