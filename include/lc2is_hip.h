/* lc2is_hip.h — C ABI of liblc2is_hip.so, the MI355X (gfx950) kernels behind the LC2IS hot path.
 *
 * The reference (AntoineBlanot/LC2IS) has no FFI layer: its hot path is PyTorch / transformers operator
 * calls inside nn.Module.forward.  Each entry point below replaces one such operator
 * call (forward or its autograd backward); the citation after "replaces:" is the reference call site
 * (paths relative to the reference root; "hf:" = transformers/models/clip/modeling_clip.py, "torch:" =
 * torch/nn).  INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers + sizes; no torch types.  All pointers are DEVICE pointers unless noted.
 *   - bf16 tensors are raw uint16 bit patterns (void* here); fp32 tensors are float*.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), re-entrant, allocates
 *     nothing, keeps no global mutable state; workspaces are caller-owned.
 *   - return value: 0 = launched; negative = refused before any launch (LC2IS_ERR_*).  Never throws.
 *   - row-major 2-D operands carry an explicit leading dimension (elements).
 */
#ifndef LC2IS_HIP_H
#define LC2IS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LC2IS_ACT_NONE 0
#define LC2IS_ACT_QUICK_GELU 1  /* x*sigmoid(1.702x), hf:activations.py:122-123                      */
#define LC2IS_ACT_RELU 2        /* F.relu default of model/decoder.py:11                                */
#define LC2IS_ACT_DQUICK_GELU 3 /* backward: acc * quick_gelu'(aux_in)                                  */
#define LC2IS_ACT_DRELU 4       /* backward: acc * (aux_in > 0)                                          */

typedef void* lc2is_stream_t; /* hipStream_t */

/* ABI / build identification: returns a static string "lc2is_hip <abi> gfx950". Host memory. */
const char* lc2is_version(void);

/* ---- dense layers ------------------------------------------------------------------------------
 * out[M,N] = epi(A[M,K] · W[N,K]^T + bias[N]) (+ resid[M,N]); K % 64 == 0, N % 4 == 0.
 * act: QUICK_GELU/RELU apply after bias and (if aux_out) store the pre-activation as bf16;
 *      DQUICK_GELU/DRELU multiply by the activation derivative at aux_in (saved pre-activation /
 *      saved relu output) — used by the dgrad of fc1 / linear1.
 * Either or both of out_bf16 / out_f32 may be given.  tile_cfg 0 = auto.
 * replaces: nn.Linear.forward at hf:CLIPAttention.forward (q/k/v/out_proj), hf:CLIPMLP.forward,
 *   torch:nn/functional.py multi_head_attention_forward in/out projections, DecoderLayer linear1/2
 *   (model/decoder.py:9-21), TextToPatch.forward (model/text_patch.py:14-19),
 *   torch.matmul(feature_v, feature_t.T) (model/model.py:50), and — with W^T shadows — their dgrads. */
int lc2is_gemm_nt_bf16(const void* A, int lda, const void* W, int ldw, const float* bias,
                       const float* resid, int ldr, const void* aux_in, int ldx, void* out_bf16, int ldo,
                       float* out_f32, int ldf, void* aux_out, int ldy, int M, int N, int K, int act,
                       int tile_cfg, lc2is_stream_t stream);

/* dW[N,K] (fp32) = dY[M,N]^T · X[M,K]  (weight gradient of out = X·W^T), reduced over M.
 * The M range is cut into `splits` slabs (workspace = splits*N*K fp32) summed by a second launch, so
 * the result is bitwise reproducible.  accumulate != 0 adds into dW instead of overwriting.
 * N % 16 == 0? no: N % 8 == 0 and K % 8 == 0.   replaces: autograd of the nn.Linear calls above. */
size_t lc2is_gemm_tn_workspace_bytes(int M, int N, int K);
int lc2is_gemm_tn_bf16(const void* dY, int ldy, const void* X, int ldx, float* dW, int ldw, int M, int N,
                       int K, int accumulate, void* workspace, size_t workspace_bytes,
                       lc2is_stream_t stream);

/* db[N] (fp32) = column sums of dY[M,N] (bias gradient). workspace >= lc2is_colsum_workspace_bytes. */
size_t lc2is_colsum_workspace_bytes(int M, int N);
int lc2is_colsum_bf16(const void* dY, int ldy, float* db, int M, int N, int accumulate, void* workspace,
                      size_t workspace_bytes, lc2is_stream_t stream);

/* ---- LayerNorm -----------------------------------------------------------------------------------
 * y = (x - mean)/sqrt(var + eps) * gamma + beta over the last dim C (C % 4 == 0, C <= 8192);
 * x fp32 [M,C] (the residual stream is kept in fp32), y bf16; mean/rstd fp32 [M] saved for backward
 * (may be NULL in inference).  gamma/beta fp32, beta may be NULL (torch 2.10 bias=False drift).
 * replaces: nn.LayerNorm.forward at hf:CLIPEncoderLayer.forward:362-383, pre_layrnorm / final_layer_norm,
 *   norm1-3 of torch TransformerDecoderLayer (model/decoder.py:9). */
int lc2is_layernorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, void* y_bf16,
                        int ldy, float* y_f32, int ldyf, float* mean, float* rstd, int M, int C, float eps,
                        lc2is_stream_t stream);

/* dx = LN'(dy) (+ dres), written as fp32 and/or bf16; dgamma/dbeta accumulated over rows through
 * `workspace` (>= lc2is_layernorm_bwd_workspace_bytes) and a second launch (deterministic).
 * dy is bf16 [M,C] (the dgrad GEMM's output) or, if dy_f32 != NULL, fp32. */
size_t lc2is_layernorm_bwd_workspace_bytes(int M, int C);
int lc2is_layernorm_bwd(const void* dy_bf16, int lddy, const float* dy_f32, int lddyf, const float* x,
                        int ldx, const float* gamma, const float* mean, const float* rstd,
                        const float* dres, int lddres, float* dx_f32, int lddx, void* dx_bf16, int lddxb,
                        float* dgamma, float* dbeta, int accumulate, int M, int C, void* workspace,
                        size_t workspace_bytes, lc2is_stream_t stream);

/* ---- attention -----------------------------------------------------------------------------------
 * O[b,s,h,:] = softmax_k( scale * Q[b,s,h,:]·K[b,k,h,:] + kbias[b,k] (+ causal) ) · V[b,k,h,:]
 * Q/K/V/O are token-major 2-D views: row (b*S + s), head h in columns [h*D,(h+1)*D), row stride ld*
 * (elements) — Q, K, V may alias one packed projection buffer.  D in {64, 96, 128}.
 * kbias: fp32 [B,Sk] additive key bias (0 = attend, -inf = masked; this is key_padding_mask /
 * attention_mask), NULL = none.  causal != 0 adds the lower-triangular mask (requires Sq == Sk).
 * lse2 (optional, fp32 [B,H,Sq]): log2-domain log-sum-exp of the scaled, masked scores, saved for backward.
 * replaces: hf:modeling_clip.py:259-277 eager_attention_forward (+ :298-335), and the attention core of
 *   torch:nn/functional.py multi_head_attention_forward used by model/decoder.py:9-21. */
int lc2is_attention_fwd(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O,
                        int ldo, float* lse2, const float* kbias, int B, int H, int Sq, int Sk, int D,
                        float scale, int causal, lc2is_stream_t stream);

/* Backward of lc2is_attention_fwd: dQ, dK, dV (bf16, same 2-D strided views as Q/K/V — they may alias one
 * packed dQKV buffer) from dO, the forward's O and lse2.  `delta` is fp32 [B,H,Sq] scratch (rowsum(dO*O)).
 * Two launches (dQ; then dK/dV), no atomics: bitwise reproducible.
 * replaces: autograd of the attention cores above (reference engine.py:100 loss.backward()). */
int lc2is_attention_bwd(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                        const void* O, int ldo, const void* dO, int lddo, void* dQ, int lddq, void* dK,
                        int lddk, void* dV, int lddv, const float* lse2, float* delta, const float* kbias,
                        int B, int H, int Sq, int Sk, int D, float scale, int causal, lc2is_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LC2IS_HIP_H */
