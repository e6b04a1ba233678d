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
};

__device__ __forceinline__ float sigmoidf_fast(float x) { return 1.0f / (1.0f + __expf(-x)); }

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_nt_kernel(GemmNtArgs p) {
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

  // ---- epilogue: lane holds C[m][n..n+3], n = 4*(lane>>4) within the 16-wide sub-tile ----
  const int act = p.act;
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = n0 + wn * WN + i * 16 + g * 4;
    if (n >= p.N) continue;
    f32x4_t bv = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (p.bias) bv = *(const f32x4_t*)(p.bias + n);
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = m0 + wm * WM + j * 16 + frow;
      if (m >= p.M) continue;
      f32x4_t v = acc[i][j] + bv;
      if (act == LC2IS_ACT_QUICK_GELU || act == LC2IS_ACT_RELU) {
        if (p.aux_out) {
          i32x2_t pk = {(int)pack_bf16x2(v[0], v[1]), (int)pack_bf16x2(v[2], v[3])};
          *(i32x2_t*)(p.aux_out + (size_t)m * p.ldy + n) = pk;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
          v[r] = (act == LC2IS_ACT_RELU) ? fmaxf(v[r], 0.f) : v[r] * sigmoidf_fast(1.702f * v[r]);
      } else if (act == LC2IS_ACT_DQUICK_GELU || act == LC2IS_ACT_DRELU) {
        const i32x2_t zk = *(const i32x2_t*)(p.aux_in + (size_t)m * p.ldx + n);
        float z[4] = {bf16_to_f32((bf16_t)(zk[0] & 0xffff)), bf16_to_f32((bf16_t)((unsigned)zk[0] >> 16)),
                      bf16_to_f32((bf16_t)(zk[1] & 0xffff)), bf16_to_f32((bf16_t)((unsigned)zk[1] >> 16))};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (act == LC2IS_ACT_DRELU) {
            v[r] = z[r] > 0.f ? v[r] : 0.f;
          } else {
            const float s = sigmoidf_fast(1.702f * z[r]);
            v[r] *= s * (1.f + 1.702f * z[r] * (1.f - s));
          }
        }
      }
      if (p.resid) v += *(const f32x4_t*)(p.resid + (size_t)m * p.ldr + n);
      if (p.out_f32) *(f32x4_t*)(p.out_f32 + (size_t)m * p.ldf + n) = v;
      if (p.out_bf16) {
        i32x2_t pk = {(int)pack_bf16x2(v[0], v[1]), (int)pack_bf16x2(v[2], v[3])};
        *(i32x2_t*)(p.out_bf16 + (size_t)m * p.ldo + n) = pk;
      }
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch_cfg(const GemmNtArgs& a, hipStream_t stream) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int LDS = 2 * (BM + BN) * 128;
  auto kern = gemm_nt_kernel<BM, BN, WAVES_M, WAVES_N>;
  static bool attr_set = false;  // idempotent; a race only repeats the same call
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return LC2IS_ERR_LAUNCH;
    attr_set = true;
  }
  const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL(kern, dim3(ntm * ntn), dim3(NT), LDS, stream, a);
  return lc2is_check_launch();
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
  if ((act == LC2IS_ACT_DQUICK_GELU || act == LC2IS_ACT_DRELU) && !aux_in) return LC2IS_ERR_NULL;
  if (act < LC2IS_ACT_NONE || act > LC2IS_ACT_DRELU) return LC2IS_ERR_UNSUPPORTED;
  // 32-bit buffer offsets: operand panels (plus one tile of overhang) must stay under 2 GiB
  if ((double)(M + 256) * lda * 2.0 >= 2147483648.0 || (double)(N + 256) * ldw * 2.0 >= 2147483648.0)
    return LC2IS_ERR_UNSUPPORTED;
  GemmNtArgs a{(const bf16_t*)A, lda, (const bf16_t*)W, ldw, bias, resid, ldr, (const bf16_t*)aux_in, ldx,
               (bf16_t*)out_bf16, ldo, out_f32, ldf, (bf16_t*)aux_out, ldy, M, N, K, act};
  int cfg = tile_cfg;
  if (cfg == 0) {
    const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (tiles128 >= 1024 && N % 128 == 0) cfg = 2;       // plenty of work: 256x128 halves W re-reads
    else if (tiles128 >= 128) cfg = 1;
    else cfg = 3;                                        // small problem: 64x64 tiles to fill the chip
  }
  switch (cfg) {
    case 1: return launch_cfg<128, 128, 2, 2>(a, stream);
    case 2: return launch_cfg<256, 128, 4, 2>(a, stream);
    case 3: return launch_cfg<64, 64, 2, 2>(a, stream);
    default: return LC2IS_ERR_UNSUPPORTED;
  }
}
