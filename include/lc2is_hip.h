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
#define LC2IS_ACT_QUICK_GELU_GRAD 5 /* forward: out = quick_gelu(z), aux_out = quick_gelu'(z) (bf16) — saves the
                                      backward's transcendentals; pair with LC2IS_ACT_MUL_AUX            */
#define LC2IS_ACT_MUL_AUX 6     /* backward: acc * aux_in (aux_in = the derivative saved by code 5)     */
#define LC2IS_ACT_GELU_ERF 7    /* exact GELU 0.5x(1+erf(x/sqrt2)), hf:activations.py "gelu" (Swin MLP) */
#define LC2IS_ACT_DGELU_ERF 8   /* backward: acc * gelu'(aux_in)                                         */
#define LC2IS_ACT_ADD_AUX 9     /* acc + bias + aux_in: the residual add of a bf16 residual stream (`x + out_proj(..)`,
                                   `x + fc2(..)` of hf CLIPEncoderLayer.forward:362-383), fp32 add, one rounding */

#define LC2IS_INTERP_BICUBIC 0  /* F.interpolate(mode="bicubic", align_corners=False), A = -0.75, border clamp */
#define LC2IS_INTERP_BILINEAR 1 /* F.interpolate(mode="bilinear", align_corners=False)                          */

typedef void* lc2is_stream_t; /* hipStream_t */

/* One fp32 master weight [N,K] (contiguous) and its bf16 shadows; see lc2is_shadow_refresh. */
typedef struct {
  const void* src; /* fp32 [N,K]                                               */
  void* dst;       /* bf16 [N, ld_dst]  row-major copy (forward GEMM operand), may be NULL */
  void* dstT;      /* bf16 [K, ld_dstT] transposed copy (dgrad GEMM operand), may be NULL  */
  int N, K, ld_dst, ld_dstT;
  int tile_start;  /* exclusive prefix sum of ceil(N/64)*ceil(K/64) over the table          */
  int flags;       /* bit 0: dst is fp32 and receives a plain copy (fused bias vectors); dstT unused     */
} lc2is_shadow_desc;

/* ABI / build identification: returns a static string "lc2is_hip <abi> gfx950". Host memory. */
const char* lc2is_version(void);

/* Compute units the GEMM tile planners may count on (0 = all 256, the default).  The large-tile kernels run one block per CU and
 * their plans are whole rounds of the CUs; a CU held by another queue's kernel for the duration (RCCL's channels while gradients
 * are reduced under the backward pass) would turn "exactly one round" into two.  With a budget n the persistent kernels launch n
 * blocks and every round count is taken over n CUs.  Process-wide.  The NT GEMM plans are bitwise equal under any budget; the
 * weight-gradient plans (lc2is_gemm_tn_bf16 / _grouped) choose their M-split count from it, i.e. the ORDER of their fp32 partial
 * sums: their results move in the last bits with the budget (each call reads the budget once and is reproducible for a given value).
 * replaces: nothing in the reference (torch DDP leaves this to the vendor GEMM library's heuristics). */
int lc2is_set_cu_budget(int ncu);
int lc2is_get_cu_budget(void);

/* ---- dense layers ------------------------------------------------------------------------------
 * out[M,N] = epi(A[M,K] · W[N,K]^T + bias[N]) (+ resid[M,N]); K % 64 == 0, N % 4 == 0.
 * act: QUICK_GELU/RELU apply after bias and (if aux_out) store the pre-activation as bf16;
 *      DQUICK_GELU/DRELU multiply by the activation derivative at aux_in (saved pre-activation /
 *      saved relu output) — used by the dgrad of fc1 / linear1.
 * Either or both of out_bf16 / out_f32 may be given.  tile_cfg 0 = auto (the plan: exact rounds of 256x256 or 256x384 tiles over
 * the chip's 256 CUs, persistent form for bf16 outputs, the <= 64 ragged rows of B x 1025-token inputs computed inside the same
 * launch); a non-zero tile_cfg forces one kernel for tests and A/B (1-3 register-staged 128x128 / 256x128 / 64x64, 4 / 6 LDS-DMA
 * 256x256 / 128x128, 13 / 15 persistent 256x256, 16 = 256x384 [N % 384 == 0, no activation, fp32-only or bf16-only output], 17 = row
 * kernel for M <= 64) and returns LC2IS_ERR_UNSUPPORTED where that kernel does not take the problem.  Every plan gives bitwise the
 * same result as tile_cfg 4.
 * replaces: nn.Linear.forward at hf:CLIPAttention.forward (q/k/v/out_proj), hf:CLIPMLP.forward,
 *   torch:nn/functional.py multi_head_attention_forward in/out projections, DecoderLayer linear1/2
 *   (model/decoder.py:9-21), TextToPatch.forward (model/text_patch.py:14-19),
 *   torch.matmul(feature_v, feature_t.T) (model/model.py:50), and — with W^T shadows — their dgrads. */
int lc2is_gemm_nt_bf16(const void* A, int lda, const void* W, int ldw, const float* bias,
                       const float* resid, int ldr, const void* aux_in, int ldx, void* out_bf16, int ldo,
                       float* out_f32, int ldf, void* aux_out, int ldy, int M, int N, int K, int act,
                       int tile_cfg, lc2is_stream_t stream);

/* The residual-stream GEMM and the LayerNorm that follows it in ONE launch (round 5): out_f32 = A.W^T + bias (+ resid) as
 * lc2is_gemm_nt_bf16 writes it, and ln_out = bf16((out_f32 - mean) * rstd * gamma + beta) per row over the N columns (mean, rstd
 * fp32 [M], may be NULL).  N = 384 or 768, M >= 256; runs on 256x384 tiles whose two column tiles exchange their row statistics
 * (two-pass variance, halves combined in a fixed order: reproducible) through `xchg` (>= lc2is_gemm_nt_ln_xchg_bytes, 8-byte
 * aligned, ALL ZERO before the first call; the kernel leaves it all zero again — one buffer per stream that may run such a launch).
 * The <= 64 ragged rows of B x 1025-token inputs are computed inside the launch and normalised by a small second launch.
 * Returns LC2IS_ERR_UNSUPPORTED for shapes it does not take: call lc2is_gemm_nt_bf16 and lc2is_layernorm_fwd instead.
 * replaces: out_proj / fc2 + residual add followed by layer_norm2 / the next layer's layer_norm1 in hf:CLIPEncoderLayer.forward
 *   (modeling_clip.py:362-383), reached from model/encoder.py:29-30. */
size_t lc2is_gemm_nt_ln_xchg_bytes(int M, int N);
int lc2is_gemm_nt_ln_bf16(const void* A, int lda, const void* W, int ldw, const float* bias, const float* resid, int ldr,
                          float* out_f32, int ldf, const float* gamma, const float* beta, float eps, void* ln_out, int ldl,
                          float* mean, float* rstd, void* xchg, size_t xchg_bytes, int M, int N, int K, lc2is_stream_t stream);

/* Strided-batched plain product, ONE launch: for b < batch, out[b][M,N] = A[b][M,K] · W[b][N,K]^T (no bias / activation);
 * stride_* are ELEMENT strides between consecutive problems (stride_a, stride_w multiples of 8, outputs multiples of 4).
 * replaces: torch.einsum('bchw,bkc->bkhw', visual, text) with per-image class embeddings (reference
 *   model/final.py:355, model/model.py:161,210, model/ftn.py:60) and, on transposed operands, its backward. */
int lc2is_gemm_nt_bf16_batched(const void* A, int lda, long stride_a, const void* W, int ldw, long stride_w,
                               void* out_bf16, int ldo, long stride_ob, float* out_f32, int ldf, long stride_of,
                               int M, int N, int K, int batch, lc2is_stream_t stream);

/* dW[N,K] (fp32) = dY[M,N]^T · X[M,K]  (weight gradient of out = X·W^T), reduced over M.
 * The M range is cut into `splits` slabs (workspace = splits*N*K fp32) summed by a second launch, so
 * the result is bitwise reproducible.  accumulate != 0 adds into dW instead of overwriting.
 * db (optional, fp32 [N]): the bias gradient colsum(dY), fused (one extra ones-fragment MFMA per tile).
 * N % 8 == 0 and K % 8 == 0.   replaces: autograd of the nn.Linear calls above. */
size_t lc2is_gemm_tn_workspace_bytes(int M, int N, int K);
int lc2is_gemm_tn_bf16(const void* dY, int ldy, const void* X, int ldx, float* dW, int ldw, float* db, int M,
                       int N, int K, int accumulate, void* workspace, size_t workspace_bytes,
                       lc2is_stream_t stream);

/* Grouped form: up to LC2IS_TN_GROUP_MAX weight gradients (one transformer layer — q, k, v, out-proj, fc1, fc2, each the
 * autograd of an nn.Linear call listed above — or a whole tower's: 12 layers = 72 problems = 1296 output tiles, scheduled as
 * full-length blocks plus a few finely split problems that fill the last round of CUs) in ONE grid and one ordered-reduce
 * launch.  More than 16 problems: the descriptor table is uploaded into the front of the workspace (one small H2D copy on
 * `stream`; such a call cannot be captured into a hipGraph).  N and K multiples of 8; a group whose N and K are all
 * multiples of 256 runs on the 256x256 LDS-DMA tiles, any other (the Swin blocks) on 128x128 tiles.  Same results contract: fp32, bitwise
 * reproducible, db (optional) = column sums of dY, `accumulate` adds to dW / db. */
#define LC2IS_TN_GROUP_MAX 128
typedef struct lc2is_tn_problem {
  const void* dY; const void* X; float* dW; float* db;
  int ldy, ldx, ldw, M, N, K, accumulate;
} lc2is_tn_problem;
size_t lc2is_gemm_tn_grouped_workspace_bytes(const lc2is_tn_problem* problems, int n);
int lc2is_gemm_tn_grouped(const lc2is_tn_problem* problems, int n, void* workspace, size_t workspace_bytes,
                          lc2is_stream_t stream);
/* A lc2is_gemm_tn_grouped call of more than 128 problems made under stream capture uploads its descriptor table through a
   memcpy node that re-reads a pinned host image at every replay; the image belongs to the graph that captured it.  This frees
   every such image (returns how many): call it once ALL graphs captured so far have been destroyed.  Per-graph ownership:
   lc2is_captured_tables_mark() before and after a capture brackets the images that capture registered, and
   lc2is_release_captured_tables_range(first, last) frees exactly those once that graph is destroyed (lc2is_amd.step.TrainStep
   does, when a captured step is released or captured again) — other live graphs keep theirs.  Slots are never reused or
   removed (a released slot stays empty), so marks stay valid across either release call; the global release is for
   tear-down only — it also frees images of graphs that are still alive.  No reference counterpart (host-side resource
   management). */
int lc2is_release_captured_tables(void);
int lc2is_captured_tables_mark(void);
int lc2is_release_captured_tables_range(int first, int last);

/* db[N] (fp32) = column sums of dY[M,N] (bias gradient). workspace >= lc2is_colsum_workspace_bytes. */
size_t lc2is_colsum_workspace_bytes(int M, int N);
int lc2is_colsum_bf16(const void* dY, int ldy, float* db, int M, int N, int accumulate, void* workspace,
                      size_t workspace_bytes, lc2is_stream_t stream);

/* ---- LayerNorm -----------------------------------------------------------------------------------
 * y = (x - mean)/sqrt(var + eps) * gamma + beta over the last dim C (C % 4 == 0, C <= 2048);
 * x fp32 [M,C] (the residual stream is kept in fp32), y bf16; mean/rstd fp32 [M] saved for backward
 * (may be NULL in inference).  gamma/beta fp32, beta may be NULL (torch 2.10 bias=False drift).
 * replaces: nn.LayerNorm.forward at hf:CLIPEncoderLayer.forward:362-383, pre_layrnorm / final_layer_norm,
 *   norm1-3 of torch TransformerDecoderLayer (model/decoder.py:9). */
int lc2is_layernorm_fwd(const void* x, int ldx, int x_is_bf16, const float* gamma, const float* beta, void* y_bf16,
                        int ldy, float* y_f32, int ldyf, float* mean, float* rstd, int M, int C, float eps,
                        lc2is_stream_t stream);

/* dx = LN'(dy) (+ dres), written as fp32 and/or bf16; dgamma/dbeta accumulated over rows through
 * `workspace` (>= lc2is_layernorm_bwd_workspace_bytes) and a second launch (deterministic).
 * dy is bf16 [M,C] (the dgrad GEMM's output) or, if dy_f32 != NULL, fp32. */
size_t lc2is_layernorm_bwd_workspace_bytes(int M, int C);
int lc2is_layernorm_bwd(const void* dy_bf16, int lddy, const float* dy_f32, int lddyf, const void* x,
                        int ldx, int x_is_bf16, const float* gamma, const float* mean, const float* rstd,
                        const void* dres, int lddres, int dres_is_bf16, float* dx_f32, int lddx, void* dx_bf16,
                        int lddxb, float* dgamma, float* dbeta, int accumulate, int M, int C, void* workspace,
                        size_t workspace_bytes, lc2is_stream_t stream);

/* Deferred parameter gradients: a lc2is_layernorm_bwd call with dgamma == dbeta == NULL leaves its per-block partial
 * sums in `workspace` — lc2is_layernorm_bwd_partials(M, C) rows of [dgamma partial (C) | dbeta partial (C)] — and
 * lc2is_ln_partials_reduce sums the partials of up to LC2IS_LN_PARTIALS_MAX such calls in ONE launch (fixed-order sums:
 * bit-identical to the per-call second launch).  Items must not share an output vector (-> LC2IS_ERR_UNSUPPORTED).
 * replaces: the weight.grad / bias.grad accumulation of every nn.LayerNorm of a tower's backward (autograd of
 *   hf:CLIPEncoderLayer.forward:362-383; 50 latency-bound 13-us launches per train step become two). */
#define LC2IS_LN_PARTIALS_MAX 64
typedef struct {
  const float* partials; /* the workspace of the lc2is_layernorm_bwd call */
  float* dgamma;         /* [C] or NULL */
  float* dbeta;          /* [C] or NULL */
  int nparts;            /* lc2is_layernorm_bwd_partials(M, C) of that call */
  int C;
  int accumulate;        /* 0: overwrite, 1: add to the vectors */
} lc2is_ln_partials;
int lc2is_layernorm_bwd_partials(int M, int C);
int lc2is_ln_partials_reduce(const lc2is_ln_partials* items, int n, lc2is_stream_t stream);

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

/* The same two operators with dropout on the attention probabilities (training mode of torch's
 * multi_head_attention_forward, dropout_p = the layer's `dropout`: torch:nn/functional.py:6206, reached from
 * PromptLayer model/decoder.py:24-28, the SR layers model/hierarchical.py:174-225 / model/decoder.py:113-134 and
 * nn.TransformerDecoderLayer at model/ftn.py:135).  No mask is stored: keep(seed, row=(b*H+h)*Sq+q, col=key) is a
 * counter-based hash evaluated in the forward and again in both backward kernels (csrc/common.h); survivors are scaled
 * by 1/(1-p).  The stream is this library's, not torch's Philox stream; lc2is_dropout_mask exports it for tests. */
int lc2is_attention_fwd_dropout(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O,
                                int ldo, float* lse2, const float* kbias, int B, int H, int Sq, int Sk, int D,
                                float scale, int causal, float p_drop, unsigned long long seed, lc2is_stream_t stream);
int lc2is_attention_bwd_dropout(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                                const void* O, int ldo, const void* dO, int lddo, void* dQ, int lddq, void* dK,
                                int lddk, void* dV, int lddv, const float* lse2, float* delta, const float* kbias,
                                int B, int H, int Sq, int Sk, int D, float scale, int causal, float p_drop,
                                unsigned long long seed, lc2is_stream_t stream);

/* ---- dropout / drop-path (training mode), mask never materialised ------------------------------------------------
 * dropout_rows_f32: y[m][c] = resid[m][c] + keep * x[m][c] / (1-p)   (resid optional; fp32 and / or bf16 output).
 *   rows_per_sample == 0: one Bernoulli(1-p) decision per element, coordinate (m, c) — nn.Dropout on a branch output
 *   (dropout1/2/3 of torch's Transformer layers) and, applied to a gradient with the forward's seed, its backward.
 *   rows_per_sample > 0: one decision per SAMPLE (row m belongs to sample m / rows_per_sample) — hf SwinDropPath
 *   (modeling_swin.py:280-302) and its backward.
 * dropout_rows_bf16: the same per-element form on bf16 (the dropout between activation and linear2).
 * dropout_mask: out[r][c] = keep(seed, r, c) as bytes — test / debug export of the decisions, never on the product path. */
int lc2is_dropout_rows_f32(const float* x, int ldx, const float* resid, int ldr, float* y32, int ldy, void* y16,
                           int ldy16, int M, int C, int rows_per_sample, float p, unsigned long long seed,
                           lc2is_stream_t stream);
int lc2is_dropout_rows_bf16(const void* x, int ldx, void* y, int ldy, int M, int C, float p, unsigned long long seed,
                            lc2is_stream_t stream);
int lc2is_dropout_mask(unsigned char* out, long rows, int cols, float p, unsigned long long seed, lc2is_stream_t stream);


/* ---- glue (all single-pass, HBM-bound) --------------------------------------------------------------
 * Refresh every bf16 weight shadow from the fp32 master copy in ONE launch: `descs` is a DEVICE array of
 * ndesc descriptors ordered by tile_start; total_tiles = sum of tile counts.  K % 4 == 0 always, N % 4 == 0
 * when dstT != NULL.  Several descriptors may target sub-blocks of one fused buffer (q/k/v -> [3C,C]). */
int lc2is_shadow_refresh(const lc2is_shadow_desc* descs, int ndesc, int total_tiles, lc2is_stream_t stream);
int lc2is_cast_f32_bf16(const float* src, int ld_src, void* dst_bf16, int ld_dst, int M, int C,
                        lc2is_stream_t stream);
int lc2is_transpose_bf16(const void* src, int ld_src, void* dst, int ld_dst, int R, int C,
                         lc2is_stream_t stream);
/* batch of `batch` such transposes in one launch; stride_* = elements between consecutive matrices. */
int lc2is_transpose_bf16_batched(const void* src, int ld_src, long stride_src, void* dst, int ld_dst, long stride_dst,
                                 int R, int C, int batch, lc2is_stream_t stream);

/* ViT patch embedding operand: out[(b*G*G + gy*G + gx)][c*p*p + i*p + j] = pixels[b][c][gy*p+i][gx*p+j]
 * (bf16, G = H / patch, trailing pixels dropped like a stride-p conv); columns [3*p*p, ld_out) are zeroed.
 * replaces: nn.Conv2d(3, C, patch, stride=patch, bias=False) im2col at hf:modeling_clip.py:202-218. */
int lc2is_patchify(const float* pixels, void* out_bf16, int ld_out, int B, int H, int W, int patch,
                   lc2is_stream_t stream);
/* x[b,0] = cls + pos[0]; x[b,1+p] = patch[b*P+p] + pos[1+p]   (hf:modeling_clip.py:211-217) and backward. */
int lc2is_vit_embed_fwd(const float* patch, int ld_patch, const float* cls, const float* pos, float* x,
                        int ldx, int B, int P, int C, lc2is_stream_t stream);
int lc2is_vit_embed_bwd(const float* dx, int ldx, float* dpos, float* dcls, void* dpatch_bf16, int ld_dpatch,
                        int B, int P, int C, int accumulate, lc2is_stream_t stream);
/* x[b*L+l] = token_embedding[ids[b,l]] + position_embedding[l]  (hf CLIPTextEmbeddings.forward) and
 * backward (dtok via fp32 atomics into a caller-zeroed/running [vocab,C] gradient). */
int lc2is_text_embed_fwd(const int64_t* ids, const float* tok, const float* pos, float* x, int ldx, int B,
                         int L, int C, int vocab, lc2is_stream_t stream);
/* dtok (zeros or the running gradient) += per-token sums in row order, no atomics: bitwise reproducible */
int lc2is_text_embed_bwd(const int64_t* ids, const float* dx, int ldx, float* dtok, float* dpos, int B, int L,
                         int C, int vocab, int accumulate, lc2is_stream_t stream);
/* dst[b, dst_off+s, :] = src[b, src_off+s, :], s < n (fp32 rows of C, optional bf16 copy): drops / re-inserts
 * the CLS token (`last_hidden_state[:, 1:, :]`, model/encoder.py:30). */
int lc2is_rows_copy_f32(const float* src, int S_src, int src_off, float* dst_f32, void* dst_bf16, int S_dst,
                        int dst_off, int B, int n, int C, lc2is_stream_t stream);
/* the same from bf16 rows (a bf16 residual stream): widened into dst_f32 and / or copied into dst_bf16 */
int lc2is_rows_copy_bf16(const void* src, int S_src, int src_off, float* dst_f32, void* dst_bf16, int S_dst,
                         int dst_off, int B, int n, int C, lc2is_stream_t stream);

/* Fused optimizer step over the flat fp32 parameter arena (n % 4 == 0).  g is multiplied by grad_scale
 * (1/world_size for DP).  SGD: torch.optim.SGD semantics (momentum_buf may be NULL); AdamW: torch.optim.AdamW.
 * replaces: optimizer.step() at engine.py:101. */
int lc2is_sgd_step(float* params, const float* grads, float* momentum_buf, size_t n, float lr, float momentum,
                   float weight_decay, float grad_scale, lc2is_stream_t stream);
int lc2is_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                     lc2is_stream_t stream);

/* ---- segmentation head tail ------------------------------------------------------------------------
 * scores_lo: fp32 channels-last [B,h,w,ld] (C valid classes, ld in {64,128,192}); output grid H = h*S,
 * W = w*S.  Computes upsample(mode) -> softmax CE against labels[B,H,W] (int64):
 *   loss_sum[0] = sum of per-pixel losses, loss_sum[1] = number of counted pixels;
 *   dscores_lo (optional, same layout) = grad_scale * U^T (softmax - onehot);
 *   scores_hi (optional) = upsampled scores as NCHW fp32 [B,C,H,W] (the reference's `outputs`).
 * S in {4, 8, 16} (every configuration of the reference): NO float atomics — the blocks' loss partials and gradient
 *   footprints go to `workspace` (>= lc2is_head_upsample_ce_workspace_bytes, 16-byte aligned; required whenever loss_sum is
 *   given) and a second launch sums them in a fixed order: loss and gradient are bitwise reproducible; loss_sum and every
 *   element of dscores_lo (all ld channels) are OVERWRITTEN, no clearing by the caller.
 * other S (multiples of 16 from 32): one launch that ADDS into loss_sum / dscores_lo with fp32 atomics (the caller clears
 *   both; results differ in the last bits from run to run); no workspace (the size query returns 0).
 * replaces: model/model.py:41-53 (bicubic x4 + TextToPatch.visual + prototype matmul, commuted), CE at
 *   engine.py:94, AuxiliaryLoss (model/loss.py:17-21, bilinear). */
size_t lc2is_head_upsample_ce_workspace_bytes(int B, int h, int w, int C, int S, int mode, int want_grad);
int lc2is_head_upsample_ce(const float* scores_lo, int ld, const int64_t* labels, float* dscores_lo,
                           float* scores_hi, float* loss_sum, int B, int h, int w, int C, int S, int mode,
                           long ignore_index, float grad_scale, void* workspace, size_t workspace_bytes,
                           lc2is_stream_t stream);
/* Transposed upsample (autograd of F.interpolate) for the unfused path: dhi NCHW fp32 [B,C,h*S,w*S] ->
 * dlo channels-last fp32 [B,h,w,ld] (columns >= C untouched). */
int lc2is_upsample_bwd_nchw(const float* dhi, float* dlo, int ld, int B, int h, int w, int C, int S, int mode,
                            lc2is_stream_t stream);
/* Plain nn.CrossEntropyLoss on NCHW fp32 logits (the unfused drop-in path): forward saves per-pixel lse,
 * backward writes dlogits = grad_scale * (*grad_scale_dev) * (softmax - onehot). */
int lc2is_ce_nchw_fwd(const float* logits, const int64_t* labels, float* lse, float* loss_sum, int B, int C,
                      long HW, long ignore_index, lc2is_stream_t stream);
int lc2is_ce_nchw_bwd(const float* logits, const int64_t* labels, const float* lse, const float* grad_scale_dev,
                      float grad_scale, float* dlogits, int B, int C, long HW, long ignore_index,
                      lc2is_stream_t stream);

/* ---- multi-scale decoder glue (BASELINE config 5; all channels-last token tensors [B, h*w, C]) --------
 * bilinear xS upsample (align_corners=False) forward / backward.
 * replaces: rearrange + F.interpolate(mode="bilinear", scale_factor=S) + rearrange at
 *   model/hierarchical.py:103-109,146-149,166-170, model/decoder.py:66-72,106-109, model/ftn.py:113,155. */
int lc2is_bilinear_up_fwd(const float* in, float* out_f32, void* out_bf16, int B, int h, int w, int C, int S,
                          lc2is_stream_t stream);
int lc2is_bilinear_up_bwd(const float* dout, float* din_f32, void* din_bf16, int B, int h, int w, int C, int S,
                          int accumulate, lc2is_stream_t stream);
/* Spatial-reduction conv operand: out[(b,y,x)][(2i+j)*C + c] = in[(b,2y+i,2x+j)][c] (bf16); scatter != 0 applies
 * the inverse map (its backward).  replaces: the im2col of Conv2d(d, d, kernel_size=2, stride=2) in
 *   SRTransformer*._sa_block (model/hierarchical.py:191,214; model/decoder.py:124). */
int lc2is_sr_gather(const void* src_bf16, void* dst_bf16, int B, int h, int w, int C, int scatter,
                    lc2is_stream_t stream);
/* Backward of the gather accumulated onto an fp32 gradient stream: dst[(b,2y+i,2x+j)][c] += src[(b,y,x)][(2i+j)*C+c]. */
int lc2is_sr_scatter_add_f32(const void* src_bf16, float* dst_f32, int B, int h, int w, int C, lc2is_stream_t stream);
/* y = x / max(||x||_2, eps) over the last dim (F.normalize, model/final.py:353-354) and its backward. */
int lc2is_l2norm_fwd(const float* x, float* y_f32, void* y_bf16, float* inv_norm, int M, int C, float eps,
                     lc2is_stream_t stream);
int lc2is_l2norm_bwd(const float* dy, const float* x, const float* inv_norm, float* dx, int M, int C, float eps,
                     lc2is_stream_t stream);
/* out = a + b (+ c) (+ d)  — torch.stack(...).sum(0) of model/hierarchical.py:128-129. */
int lc2is_add_n(const float* a, const float* b, const float* c, const float* d, float* out_f32, void* out_bf16,
                size_t n, lc2is_stream_t stream);

/* ---- Swin backbone (model/encoder.py:121-131 -> hf:models/swin/modeling_swin.py) ---------------------------------
 * rows_gather: dst[r][0:cols] = (map[r] >= 0 ? src[map[r]][0:cols] : 0) (+ add[r][0:cols]); src/dst fp32 or bf16
 *   (flags), add fp32.  With host-built index maps this is pad + cyclic shift + window partition
 *   (modeling_swin.py:546-550), window reverse + un-shift + un-pad + residual add (:558-567), the patch-merging 2x2
 *   concat (:318-321) and each of their backwards (inverse maps). */
int lc2is_rows_gather(const void* src, int ld_src, int src_bf16, void* dst, int ld_dst, int dst_bf16, const int* map,
                      const float* add, int ld_add, int rows, int cols, lc2is_stream_t stream);
/* Window attention (modeling_swin.py:373-398,428-465): qkv [nwin*S, 3C] bf16 (q | k | v, head h at columns 32h),
 *   S = ws*ws <= 64, head_dim 32, bias [nH,S,S] fp32 = relative position bias; for shift > 0 the cyclic-shift region
 *   mask (-100 across regions, :584-607) is derived from the window index (win_per_img windows per image, nwx per
 *   row, padded grid Hp x Wp).  out [nwin*S, C] bf16, lse [nwin,nH,S].  Backward writes dqkv and the bias gradient
 *   dbias [nH,S,S] (summed over windows in a fixed order; workspace from the _workspace_bytes call). */
int lc2is_swin_attn_fwd(const void* qkv, int ld, void* out, int ldo, float* lse, const float* bias, int nwin,
                        int win_per_img, int nwx, int Hp, int Wp, int ws, int shift, int nH, int C, float scale,
                        lc2is_stream_t stream);
size_t lc2is_swin_attn_bwd_workspace_bytes(int nwin, int ws, int nH);
int lc2is_swin_attn_bwd(const void* qkv, int ld, const void* o, int ld_o, const void* dout, int lddo, const float* lse,
                        const float* bias, void* dqkv, int lddq, float* dbias, int accumulate_dbias, int nwin,
                        int win_per_img, int nwx, int Hp, int Wp, int ws, int shift, int nH, int C, float scale,
                        void* workspace, size_t workspace_bytes, lc2is_stream_t stream);

/* gradient of the relative_position_bias_table [T = (2 ws - 1)^2, nH] from dbias [nH, S*S]: dtable[t][h] = sum of
 * dbias[h][p] over the pairs p = positions[offsets[t] .. offsets[t+1]) (the inverse of relative_position_index,
 * modeling_swin.py:350-383), added in list order.  replaces: autograd of `table[index]` (index_put_ with accumulate). */
int lc2is_swin_bias_table_grad(const float* dbias, const int* offsets, const int* positions, float* dtable, int nH, int SS,
                               int T, int accumulate, lc2is_stream_t stream);

/* ---- preprocessing in front of the path (evaluate.py:58-61, data/collator.py:82-91; Pillow inside transformers'
 * CLIPFeatureExtractor) — byte / integer work, bit-exact against Pillow --------------------------------------------
 * resample_u8: one separable 8-bit pass of PIL's ImagingResample over an HWC uint8 image along `axis` (1 = width,
 *   0 = height): out = clip8((1<<21 + sum_t in[first+t] * kk[o][t]) >> 22) with bounds[o] = (first, count) and the
 *   22-bit fixed-point coefficients built on the host as Resample.c precompute_coeffs / normalize_coeffs_8bpc do.
 * gather2d_u8: dst[y][x][c] = src[yi[y]][xi[x]][c] (nearest resize, Geometry.c ImagingScaleAffine index vectors).
 * crop_lut: S x S crop at (top,left) + 256-entry lookup: uint8 HWC -> float32 CHW via lut_f32[C][256] (x/255 and
 *   (x-mean)/std folded in with the reference's float ops) and / or channel 0 -> int64 via lut_i64[256] (labels). */
int lc2is_resample_u8(const void* src_u8, int H, int W, int C, void* dst_u8, int out_size, int axis, const int* bounds,
                      const int* kk, int ksize, lc2is_stream_t stream);
int lc2is_gather2d_u8(const void* src_u8, int H, int W, int C, void* dst_u8, int out_h, int out_w, const int* yi,
                      const int* xi, lc2is_stream_t stream);
int lc2is_crop_lut(const void* src_u8, int H, int W, int C, int top, int left, int S, const float* lut_f32,
                   float* dst_f32, const int64_t* lut_i64, int64_t* dst_i64, lc2is_stream_t stream);

/* ---- remaining losses (model/loss.py) and the parity metric (metrics.py) on channels-last scores ----------
 * rows_ce: softmax-CE over the K contiguous classes of each of M rows: loss_sum[0] += sum of per-row losses,
 *   lse[M] (optional), dx (optional) (+)= grad_scale * (softmax - onehot).  ContrastiveLoss.loss_visual
 *   (model/loss.py:59) — also usable for any [M,K] logits.
 * cols_ce: ContrastiveLoss.loss_textual (model/loss.py:58): x viewed [B,H,W,K], log-softmax over H (dim 1 — what
 *   nn.CrossEntropyLoss does with the reference's one-hot float targets), loss_sum[0] += sum over (b,w,k) columns;
 *   dx += grad_scale * d/dx. */
int lc2is_rows_ce(const float* x, const int64_t* labels, float* lse, float* loss_sum, float* dx, float grad_scale,
                  int M, int K, int accumulate_dx, lc2is_stream_t stream);
int lc2is_cols_ce(const float* x, const int64_t* labels, float* loss_sum, float* dx, float grad_scale, int B, int H,
                  int W, int K, lc2is_stream_t stream);
/* NPairLoss.forward before its reduction (model/loss.py:30-35): res[i] = sum_p pos_ip / (pos_ip + sum_q neg_iq). */
int lc2is_npair(const float* x, const float* x_pos, const float* x_neg, float* res, int n, int n_pos, int n_neg, int d,
                lc2is_stream_t stream);
/* Its backward: dres [n] = gradient of res; workspace = n * (n_pos + 1) floats.  Fixed summation order. */
int lc2is_npair_bwd(const float* x, const float* x_pos, const float* x_neg, const float* dres, float* dx, float* dx_pos,
                    float* dx_neg, float* workspace, int n, int n_pos, int n_neg, int d, lc2is_stream_t stream);
/* compute_mIOU's confusion counts (metrics.py:82-102): counts[b] = {intersection[K], predicted[K], labelled[K]} (int32,
 * caller-zeroed) from NCHW scores at the upsampled size and the nearest-x S labels. */
int lc2is_miou_counts(const float* scores_hi, const int64_t* labels_lo, int* counts, int B, int K, int H, int W, int S,
                      lc2is_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LC2IS_HIP_H */
