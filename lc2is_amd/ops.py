"""Thin tensor-level wrappers over the C ABI (``include/lc2is_hip.h``).

Each function validates dtype / device / layout (raising ``RuntimeError`` like the reference's torch calls
would), takes raw device pointers + leading dimensions from the tensors and launches on the caller's
current HIP stream.  Outputs and workspaces are allocated here with ``torch.empty`` (PyTorch is the
allocator, nothing else).  There is no CPU path: a non-CUDA tensor is an error.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib

ACT_NONE, ACT_QUICK_GELU, ACT_RELU, ACT_DQUICK_GELU, ACT_DRELU, ACT_QUICK_GELU_GRAD, ACT_MUL_AUX = 0, 1, 2, 3, 4, 5, 6
ACT_GELU_ERF, ACT_DGELU_ERF, ACT_ADD_AUX = 7, 8, 9

_P, _I, _F, _Z, _L, _U64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_long, C.c_ulonglong
_ARGTYPES = {
    "lc2is_gemm_nt_bf16": [_P, _I, _P, _I, _P, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "lc2is_gemm_nt_ln_xchg_bytes": [_I, _I],
    "lc2is_gemm_nt_ln_bf16": [_P, _I, _P, _I, _P, _P, _I, _P, _I, _P, _P, _F, _P, _I, _P, _P, _P, _Z, _I, _I, _I, _P],
    "lc2is_gemm_nt_bf16_batched": [_P, _I, _L, _P, _I, _L, _P, _I, _L, _P, _I, _L, _I, _I, _I, _I, _P],
    "lc2is_transpose_bf16_batched": [_P, _I, _L, _P, _I, _L, _I, _I, _I, _P],
    "lc2is_gemm_tn_workspace_bytes": [_I, _I, _I],
    "lc2is_gemm_tn_bf16": [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _Z, _P],
    "lc2is_colsum_workspace_bytes": [_I, _I],
    "lc2is_colsum_bf16": [_P, _I, _P, _I, _I, _I, _P, _Z, _P],
    "lc2is_layernorm_fwd": [_P, _I, _I, _P, _P, _P, _I, _P, _I, _P, _P, _I, _I, _F, _P],
    "lc2is_layernorm_bwd_partials": [_I, _I],
    "lc2is_ln_partials_reduce": [_P, _I, _P],
    "lc2is_layernorm_bwd_workspace_bytes": [_I, _I],
    "lc2is_layernorm_bwd": [_P, _I, _P, _I, _P, _I, _I, _P, _P, _P, _P, _I, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I,
                            _P, _Z, _P],
    "lc2is_attention_fwd": [_P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "lc2is_attention_bwd": [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P,
                            _I, _I, _I, _I, _I, _F, _I, _P],
    "lc2is_attention_fwd_dropout": [_P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _F, _I, _F, _U64, _P],
    "lc2is_attention_bwd_dropout": [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P,
                                    _I, _I, _I, _I, _I, _F, _I, _F, _U64, _P],
    "lc2is_dropout_rows_f32": [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _F, _U64, _P],
    "lc2is_dropout_rows_bf16": [_P, _I, _P, _I, _I, _I, _F, _U64, _P],
    "lc2is_dropout_mask": [_P, _L, _I, _F, _U64, _P],
    "lc2is_shadow_refresh": [_P, _I, _I, _P],
    "lc2is_set_cu_budget": [_I],
    "lc2is_get_cu_budget": [],
    "lc2is_cast_f32_bf16": [_P, _I, _P, _I, _I, _I, _P],
    "lc2is_transpose_bf16": [_P, _I, _P, _I, _I, _I, _P],
    "lc2is_patchify": [_P, _P, _I, _I, _I, _I, _I, _P],
    "lc2is_vit_embed_fwd": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _P],
    "lc2is_vit_embed_bwd": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "lc2is_text_embed_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "lc2is_text_embed_bwd": [_P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P],
    "lc2is_rows_copy_bf16": [_P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _P],
    "lc2is_rows_copy_f32": [_P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _P],
    "lc2is_sgd_step": [_P, _P, _P, _Z, _F, _F, _F, _F, _P],
    "lc2is_adamw_step": [_P, _P, _P, _P, _Z, _F, _F, _F, _F, _F, _I, _F, _P],
    "lc2is_head_upsample_ce_workspace_bytes": [_I, _I, _I, _I, _I, _I, _I],
    "lc2is_head_upsample_ce": [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, C.c_long, _F, _P, _Z, _P],
    "lc2is_ce_nchw_fwd": [_P, _P, _P, _P, _I, _I, C.c_long, C.c_long, _P],
    "lc2is_ce_nchw_bwd": [_P, _P, _P, _P, _F, _P, _I, _I, C.c_long, C.c_long, _P],
    "lc2is_upsample_bwd_nchw": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "lc2is_bilinear_up_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "lc2is_bilinear_up_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "lc2is_sr_gather": [_P, _P, _I, _I, _I, _I, _I, _P],
    "lc2is_sr_scatter_add_f32": [_P, _P, _I, _I, _I, _I, _P],
    "lc2is_l2norm_fwd": [_P, _P, _P, _P, _I, _I, _F, _P],
    "lc2is_l2norm_bwd": [_P, _P, _P, _P, _I, _I, _F, _P],
    "lc2is_add_n": [_P, _P, _P, _P, _P, _P, _Z, _P],
    "lc2is_rows_ce": [_P, _P, _P, _P, _P, _F, _I, _I, _I, _P],
    "lc2is_cols_ce": [_P, _P, _P, _P, _F, _I, _I, _I, _I, _P],
    "lc2is_npair": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "lc2is_miou_counts": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "lc2is_npair_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "lc2is_gemm_tn_grouped_workspace_bytes": [_P, _I],
    "lc2is_gemm_tn_grouped": [_P, _I, _P, _Z, _P],
    "lc2is_release_captured_tables": [],
    "lc2is_captured_tables_mark": [],
    "lc2is_release_captured_tables_range": [_I, _I],
    "lc2is_rows_gather": [_P, _I, _I, _P, _I, _I, _P, _P, _I, _I, _I, _P],
    "lc2is_resample_u8": [_P, _I, _I, _I, _P, _I, _I, _P, _P, _I, _P],
    "lc2is_gather2d_u8": [_P, _I, _I, _I, _P, _I, _I, _P, _P, _P],
    "lc2is_crop_lut": [_P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "lc2is_swin_attn_fwd": [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P],
    "lc2is_swin_bias_table_grad": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "lc2is_swin_attn_bwd_workspace_bytes": [_I, _I, _I],
    "lc2is_swin_attn_bwd": [_P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F,
                            _P, _Z, _P],
}
_bound = {}


def _fn(name: str):
    f = _bound.get(name)
    if f is None:
        f = getattr(_lib.load(), name)
        if name in _ARGTYPES:
            f.argtypes = _ARGTYPES[name]
        if name.endswith("_workspace_bytes"):
            f.restype = C.c_size_t
        _bound[name] = f
    return f


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor | None, dtype, name: str, ndim: int | None = 2):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError(f"lc2is_amd: {name} must be a CUDA(HIP) tensor; there is no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"lc2is_amd: {name} must be {dtype}, got {t.dtype}")
    if ndim is not None and t.dim() != ndim:
        raise RuntimeError(f"lc2is_amd: {name} must be {ndim}-D, got shape {tuple(t.shape)}")
    if t.dim() >= 1 and t.stride(-1) != 1:
        raise RuntimeError(f"lc2is_amd: {name} must have unit stride in its last dimension")


def _ptr(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


def _ld(t: torch.Tensor | None) -> int:
    return 0 if t is None else (t.stride(0) if t.dim() == 2 else t.shape[-1])


def _out(spec, shape, dtype, dev):
    if spec is True:
        return torch.empty(shape, dtype=dtype, device=dev)
    if spec is None or spec is False:
        return None
    return spec



_ws_cache: dict = {}


def workspace(nbytes: int, device, tag: str = "default") -> torch.Tensor:
    """Grow-only scratch buffer per (device, stream, tag); stream-ordered reuse is safe because every
    consumer of a workspace is enqueued on the same stream before the next producer."""
    key = (str(device), _stream(), tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def gemm_nt(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None = None, *, act: int = ACT_NONE,
            resid: torch.Tensor | None = None, aux_in: torch.Tensor | None = None,
            out_bf16: torch.Tensor | bool | None = True, out_f32: torch.Tensor | bool | None = None,
            aux_out: torch.Tensor | bool | None = None, tile_cfg: int = 0):
    """out = epi(a @ w.T + bias) (+ resid).  a [M,K] bf16, w [N,K] bf16, bias fp32 [N], resid fp32 [M,N] — or bf16 (a bf16
    residual stream: the add runs as the ACT_ADD_AUX epilogue, needs act == ACT_NONE and no other aux_in).
    ``out_bf16`` / ``out_f32`` / ``aux_out``: True = allocate, tensor = write into it, None/False = skip.
    Returns (out_bf16, out_f32, aux_out) with None for skipped outputs."""
    _chk(a, torch.bfloat16, "a"); _chk(w, torch.bfloat16, "w")
    if resid is not None and resid.dtype == torch.bfloat16:
        if act != ACT_NONE or aux_in is not None:
            raise RuntimeError("lc2is_amd.gemm_nt: a bf16 residual needs act == ACT_NONE and no aux_in")
        act, aux_in, resid = ACT_ADD_AUX, resid, None
    _chk(bias, torch.float32, "bias", 1); _chk(resid, torch.float32, "resid"); _chk(aux_in, torch.bfloat16, "aux_in")
    M, K = a.shape
    N, K2 = w.shape
    if K2 != K:
        raise RuntimeError(f"lc2is_amd.gemm_nt: K mismatch {K} vs {K2}")
    if bias is not None and bias.numel() != N:
        raise RuntimeError("lc2is_amd.gemm_nt: bias length != N")
    dev = a.device
    ob = _out(out_bf16, (M, N), torch.bfloat16, dev)
    of = _out(out_f32, (M, N), torch.float32, dev)
    ao = _out(aux_out, (M, N), torch.bfloat16, dev)
    _chk(ob, torch.bfloat16, "out_bf16"); _chk(of, torch.float32, "out_f32"); _chk(ao, torch.bfloat16, "aux_out")
    rc = _fn("lc2is_gemm_nt_bf16")(_ptr(a), _ld(a), _ptr(w), _ld(w), _ptr(bias), _ptr(resid), _ld(resid),
                                   _ptr(aux_in), _ld(aux_in), _ptr(ob), _ld(ob), _ptr(of), _ld(of), _ptr(ao),
                                   _ld(ao), M, N, K, act, tile_cfg, _stream())
    _lib.check(rc, f"gemm_nt M={M} N={N} K={K}")
    return ob, of, ao


_xchg_cache = {}
# the GEMM + LayerNorm launch (round 5): LC2IS_LN_FUSE=0 keeps the two launches everywhere; rows from which the fused form is taken
_LN_FUSE = os.environ.get("LC2IS_LN_FUSE", "1") != "0"
# 0 = where the 256x384 tiles fill the chip anyway: 224 tiles (7/8 of a round of the 256 CUs, the large-tile rule of lc2is_gemm_nt_bf16) —
# M >= 28 672 at N = 768; below that the unfused plan's smaller tiles keep more CUs busy than 2 x M / 256 blocks would
_LN_FUSE_MIN_ROWS = int(os.environ.get("LC2IS_LN_FUSE_MIN_ROWS", "0"))


def gemm_nt_ln_ok(M: int, N: int, K: int) -> bool:
    """Should this problem take gemm_nt_ln (else: gemm_nt + layernorm_fwd)?"""
    if not (_LN_FUSE and N in (384, 768) and K % 64 == 0 and M >= 256 and M % 256 <= 64):
        return False
    if _LN_FUSE_MIN_ROWS > 0:
        return M >= _LN_FUSE_MIN_ROWS
    return (M // 256) * (N // 384) >= 224


def gemm_nt_ln(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None, resid: torch.Tensor | None, gamma: torch.Tensor,
               beta: torch.Tensor | None, eps: float = 1e-5, *, save_stats: bool = True,
               out_f32: torch.Tensor | None = None, ln_out: torch.Tensor | None = None):
    """x = a @ w.T + bias (+ resid) in fp32 AND h = bf16(LayerNorm(x) * gamma + beta) in ONE launch (256x384 tiles, the two column
    tiles of a row tile exchange row statistics).  a [M,K] bf16, w [N,K] bf16 with N in (384, 768), resid fp32 [M,N].
    Returns (x_f32, h_bf16, mean, rstd) — what gemm_nt(..., out_f32=True) followed by layernorm_fwd returns."""
    _chk(a, torch.bfloat16, "a"); _chk(w, torch.bfloat16, "w"); _chk(bias, torch.float32, "bias", 1)
    _chk(resid, torch.float32, "resid"); _chk(gamma, torch.float32, "gamma", 1); _chk(beta, torch.float32, "beta", 1)
    M, K = a.shape
    N, K2 = w.shape
    if K2 != K or gamma.numel() != N or (bias is not None and bias.numel() != N) or (beta is not None and beta.numel() != N):
        raise RuntimeError("lc2is_amd.gemm_nt_ln: shape mismatch")
    dev = a.device
    xf = _out(out_f32 if out_f32 is not None else True, (M, N), torch.float32, dev)
    h = _out(ln_out if ln_out is not None else True, (M, N), torch.bfloat16, dev)
    mean = torch.empty((M,), dtype=torch.float32, device=dev) if save_stats else None
    rstd = torch.empty((M,), dtype=torch.float32, device=dev) if save_stats else None
    nb = _fn("lc2is_gemm_nt_ln_xchg_bytes")(M, N)
    xc = None
    if nb:
        key = (str(dev), _stream())
        xc = _xchg_cache.get(key)
        if xc is None or xc.numel() < nb:
            xc = torch.zeros(max(int(nb), 1 << 21), dtype=torch.uint8, device=dev)   # all zero once; every launch leaves it all zero
            _xchg_cache[key] = xc
    rc = _fn("lc2is_gemm_nt_ln_bf16")(_ptr(a), _ld(a), _ptr(w), _ld(w), _ptr(bias), _ptr(resid), _ld(resid), _ptr(xf), _ld(xf),
                                      _ptr(gamma), _ptr(beta), float(eps), _ptr(h), _ld(h), _ptr(mean), _ptr(rstd), _ptr(xc),
                                      xc.numel() if xc is not None else 0, M, N, K, _stream())
    _lib.check(rc, f"gemm_nt_ln M={M} N={N} K={K}")
    return xf, h, mean, rstd


def gemm_nt_batched(a: torch.Tensor, w: torch.Tensor, *, out_bf16: torch.Tensor | bool | None = None,
                    out_f32: torch.Tensor | bool | None = True):
    """out[b] = a[b] @ w[b].T for every b in ONE launch.  a [B,M,K] bf16, w [B,N,K] bf16 (unit inner stride; any row /
    batch strides that are multiples of 8 elements).  Returns (out_bf16 [B,M,N] or None, out_f32 [B,M,N] or None)."""
    _chk(a, torch.bfloat16, "a", 3); _chk(w, torch.bfloat16, "w", 3)
    Bn, M, K = a.shape
    if w.shape[0] != Bn or w.shape[2] != K:
        raise RuntimeError(f"lc2is_amd.gemm_nt_batched: shape mismatch {tuple(a.shape)} vs {tuple(w.shape)}")
    N = w.shape[1]
    ob = _out(out_bf16, (Bn, M, N), torch.bfloat16, a.device)
    of = _out(out_f32, (Bn, M, N), torch.float32, a.device)
    _chk(ob, torch.bfloat16, "out_bf16", 3); _chk(of, torch.float32, "out_f32", 3)
    st = lambda t: (0, 0) if t is None else (t.stride(1), t.stride(0))  # noqa: E731
    rc = _fn("lc2is_gemm_nt_bf16_batched")(_ptr(a), a.stride(1), a.stride(0), _ptr(w), w.stride(1), w.stride(0),
                                           _ptr(ob), *st(ob), _ptr(of), *st(of), M, N, K, Bn, _stream())
    _lib.check(rc, f"gemm_nt_batched B={Bn} M={M} N={N} K={K}")
    return ob, of


def transpose_bf16_batched(x: torch.Tensor) -> torch.Tensor:
    """x [B,R,C] bf16 -> [B,C,R] contiguous, one launch."""
    _chk(x, torch.bfloat16, "x", 3)
    Bn, R, Cc = x.shape
    out = torch.empty((Bn, Cc, R), dtype=torch.bfloat16, device=x.device)
    rc = _fn("lc2is_transpose_bf16_batched")(_ptr(x), x.stride(1), x.stride(0), _ptr(out), R, Cc * R, R, Cc, Bn, _stream())
    _lib.check(rc, f"transpose_bf16_batched B={Bn} R={R} C={Cc}")
    return out


def gemm_tn(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor | None = None, accumulate: bool = False,
            db: torch.Tensor | None = None):
    """dw[N,K] (fp32) = dy[M,N]^T @ x[M,K]; optional fused bias gradient db[N] = dy.sum(0) (same accumulate flag)."""
    _chk(dy, torch.bfloat16, "dy"); _chk(x, torch.bfloat16, "x")
    M, N = dy.shape
    M2, K = x.shape
    if M2 != M:
        raise RuntimeError("lc2is_amd.gemm_tn: M mismatch")
    if dw is None:
        dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
        accumulate = False
    _chk(dw, torch.float32, "dw"); _chk(db, torch.float32, "db", 1)
    if db is not None and db.numel() != N:
        raise RuntimeError("lc2is_amd.gemm_tn: db length != N")
    nbytes = _fn("lc2is_gemm_tn_workspace_bytes")(M, N, K)
    ws = workspace(nbytes, dy.device, "gemm_tn")
    rc = _fn("lc2is_gemm_tn_bf16")(_ptr(dy), _ld(dy), _ptr(x), _ld(x), _ptr(dw), _ld(dw), _ptr(db), M, N, K,
                                   int(accumulate), _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, f"gemm_tn M={M} N={N} K={K}")
    return dw


class TnProblem(C.Structure):
    _fields_ = [("dY", C.c_void_p), ("X", C.c_void_p), ("dW", C.c_void_p), ("db", C.c_void_p), ("ldy", C.c_int),
                ("ldx", C.c_int), ("ldw", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("accumulate", C.c_int)]


GROUP_MAX = 128        # LC2IS_TN_GROUP_MAX


# rows (tokens) from which a weight gradient joins a grouped launch: the text tower (151 prompts x 16 tokens = 2416 rows, 72
# problems of 4..16 tiles) stays 76 small launches on its side stream: grouping them (1024) measured neutral on the step
_GROUP_MIN_ROWS = int(os.environ.get("LC2IS_GROUP_MIN_ROWS", "4096"))


def gemm_tn_groupable(dy: torch.Tensor, x: torch.Tensor) -> bool:
    return dy.shape[1] % 8 == 0 and x.shape[1] % 8 == 0 and dy.shape[0] >= _GROUP_MIN_ROWS


def gemm_tn_grouped(problems):
    """problems: list of (dy [M,N] bf16, x [M,K] bf16, dw [N,K] fp32, db [N] fp32 or None, accumulate) with N, K
    multiples of 8 (check with gemm_tn_groupable; groups whose N, K are all multiples of 256 take the 256x256 LDS-DMA tiles) — the weight gradients of one layer, or of a whole tower, in one grid."""
    if not 1 <= len(problems) <= GROUP_MAX:
        raise RuntimeError(f"lc2is_amd.gemm_tn_grouped: 1..{GROUP_MAX} problems per launch")
    arr = (TnProblem * len(problems))()
    for i, (dy, x, dw, db, acc) in enumerate(problems):
        _chk(dy, torch.bfloat16, "dy"); _chk(x, torch.bfloat16, "x"); _chk(dw, torch.float32, "dw"); _chk(db, torch.float32, "db", 1)
        M, N = dy.shape
        if x.shape[0] != M or tuple(dw.shape) != (N, x.shape[1]) or (db is not None and db.numel() != N):
            raise RuntimeError("lc2is_amd.gemm_tn_grouped: shape mismatch")
        arr[i] = TnProblem(_ptr(dy), _ptr(x), _ptr(dw), _ptr(db), _ld(dy), _ld(x), _ld(dw), M, N, x.shape[1], int(acc))
    nbytes = _fn("lc2is_gemm_tn_grouped_workspace_bytes")(arr, len(problems))
    ws = workspace(nbytes, problems[0][0].device, "gemm_tn_grouped")
    rc = _fn("lc2is_gemm_tn_grouped")(arr, len(problems), _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, f"gemm_tn_grouped n={len(problems)}")


def release_captured_tables() -> int:
    """Free the pinned descriptor-table images that captured grouped weight-gradient launches left behind (call after the
    graphs that captured them are destroyed).  Returns the number freed."""
    f = _fn("lc2is_release_captured_tables")
    f.restype = C.c_int
    return int(f())


def captured_tables_mark() -> int:
    """Number of captured descriptor-table images registered so far (bracket a capture with two marks)."""
    return int(_fn("lc2is_captured_tables_mark")())


def release_captured_tables_range(first: int, last: int) -> int:
    """Free the images registered between two marks (the ones ONE captured graph owns), after that graph is destroyed."""
    return int(_fn("lc2is_release_captured_tables_range")(int(first), int(last)))


def colsum(dy: torch.Tensor, db: torch.Tensor | None = None, accumulate: bool = False):
    """db[N] (fp32) = dy[M,N].sum(0)."""
    _chk(dy, torch.bfloat16, "dy")
    M, N = dy.shape
    if db is None:
        db = torch.empty((N,), dtype=torch.float32, device=dy.device)
        accumulate = False
    _chk(db, torch.float32, "db", 1)
    nbytes = _fn("lc2is_colsum_workspace_bytes")(M, N)
    ws = workspace(nbytes, dy.device, "colsum")
    rc = _fn("lc2is_colsum_bf16")(_ptr(dy), _ld(dy), _ptr(db), M, N, int(accumulate), _ptr(ws), ws.numel(),
                                  _stream())
    _lib.check(rc, f"colsum M={M} N={N}")
    return db


def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor | None, eps: float = 1e-5, *,
                  save_stats: bool = True, out_bf16: torch.Tensor | bool | None = True,
                  out_f32: torch.Tensor | bool | None = None):
    """x fp32 or bf16 [M,C] (the residual stream) -> (y_bf16, y_f32, mean, rstd)."""
    xb = x.dtype == torch.bfloat16
    _chk(x, torch.bfloat16 if xb else torch.float32, "x"); _chk(gamma, torch.float32, "gamma", 1); _chk(beta, torch.float32, "beta", 1)
    M, Cc = x.shape
    dev = x.device
    yb = _out(out_bf16, (M, Cc), torch.bfloat16, dev)
    yf = _out(out_f32, (M, Cc), torch.float32, dev)
    mean = torch.empty((M,), dtype=torch.float32, device=dev) if save_stats else None
    rstd = torch.empty((M,), dtype=torch.float32, device=dev) if save_stats else None
    rc = _fn("lc2is_layernorm_fwd")(_ptr(x), _ld(x), int(xb), _ptr(gamma), _ptr(beta), _ptr(yb), _ld(yb), _ptr(yf),
                                    _ld(yf), _ptr(mean), _ptr(rstd), M, Cc, float(eps), _stream())
    _lib.check(rc, f"layernorm_fwd M={M} C={Cc}")
    return yb, yf, mean, rstd


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, mean: torch.Tensor,
                  rstd: torch.Tensor, *, dres: torch.Tensor | None = None, dgamma: torch.Tensor | None = None,
                  dbeta: torch.Tensor | None = None, accumulate: bool = False, want_f32: bool = True,
                  want_bf16: bool = True, need_param_grads: bool = True):
    """Returns (dx_f32, dx_bf16, dgamma, dbeta).  dy, x (the residual stream the forward normalised) and dres (the gradient
    arriving over the residual path) are each bf16 or fp32 [M,C]."""
    M, Cc = x.shape
    dev = x.device
    dyb = dy if dy.dtype == torch.bfloat16 else None
    dyf = dy if dy.dtype == torch.float32 else None
    if dyb is None and dyf is None:
        raise RuntimeError("lc2is_amd.layernorm_bwd: dy must be bf16 or fp32")
    xb = x.dtype == torch.bfloat16
    rb = dres is not None and dres.dtype == torch.bfloat16
    _chk(dyb, torch.bfloat16, "dy"); _chk(dyf, torch.float32, "dy"); _chk(x, torch.bfloat16 if xb else torch.float32, "x")
    _chk(dres, torch.bfloat16 if rb else torch.float32, "dres")
    dxf = torch.empty((M, Cc), dtype=torch.float32, device=dev) if want_f32 else None
    dxb = torch.empty((M, Cc), dtype=torch.bfloat16, device=dev) if want_bf16 else None
    if need_param_grads:
        if dgamma is None:
            dgamma = torch.empty((Cc,), dtype=torch.float32, device=dev)
            accumulate = False
            if dbeta is None:
                dbeta = torch.empty((Cc,), dtype=torch.float32, device=dev)
    nbytes = _fn("lc2is_layernorm_bwd_workspace_bytes")(M, Cc)
    defer = _ln_defer if (need_param_grads and (dgamma is not None or dbeta is not None)) else None
    if defer is not None:
        # parameter gradients deferred (ln_defer_begin/flush): the partial sums stay in a buffer of their own until the
        # grouped reduce at the end of the layer group
        ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)
        defer.append((ws, _fn("lc2is_layernorm_bwd_partials")(M, Cc), Cc, dgamma, dbeta, bool(accumulate)))
        pg, pb = None, None
    else:
        ws = workspace(nbytes, dev, "ln_bwd")
        pg, pb = dgamma, dbeta
    rc = _fn("lc2is_layernorm_bwd")(_ptr(dyb), _ld(dyb), _ptr(dyf), _ld(dyf), _ptr(x), _ld(x), int(xb), _ptr(gamma),
                                    _ptr(mean), _ptr(rstd), _ptr(dres), _ld(dres), int(rb), _ptr(dxf), _ld(dxf),
                                    _ptr(dxb), _ld(dxb), _ptr(pg), _ptr(pb), int(accumulate), M, Cc,
                                    _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, f"layernorm_bwd M={M} C={Cc}")
    return dxf, dxb, dgamma, dbeta


class LnPartials(C.Structure):
    _fields_ = [("partials", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("nparts", C.c_int),
                ("C", C.c_int), ("accumulate", C.c_int)]


LN_PARTIALS_MAX = 64   # LC2IS_LN_PARTIALS_MAX
_LN_DEFER = os.environ.get("LC2IS_LN_DEFER", "1") != "0"   # A/B switch: 0 = every layernorm_bwd reduces its own partials
_ln_defer = None       # list of deferred (ws, nparts, C, dgamma, dbeta, accumulate) while a deferral scope is open


def ln_defer_begin():
    """Open a deferral scope: layernorm_bwd calls that are handed dgamma / dbeta buffers keep their partial sums and the
    reductions leave together at ln_defer_flush (nn.base.WgradBatch opens / flushes one with the weight gradients).
    Returns the previous scope (pass it to ln_defer_end)."""
    global _ln_defer
    prev, _ln_defer = _ln_defer, ([] if _LN_DEFER else None)
    return prev


def ln_defer_flush():
    """One grouped launch per <= LN_PARTIALS_MAX deferred calls (two calls that write the same vector are never put in
    the same launch: the second waits for the next one)."""
    global _ln_defer
    items = _ln_defer
    if not items:
        return
    _ln_defer = []
    while items:
        chunk, rest, seen = [], [], set()
        for it in items:
            outs = {t.data_ptr() for t in (it[3], it[4]) if t is not None}
            if len(chunk) < LN_PARTIALS_MAX and not (outs & seen) and not rest:
                chunk.append(it); seen |= outs
            else:
                rest.append(it)
        arr = (LnPartials * len(chunk))()
        for i, (ws, nparts, Cc, dg, db, acc) in enumerate(chunk):
            _chk(dg, torch.float32, "dgamma", 1); _chk(db, torch.float32, "dbeta", 1)
            arr[i] = LnPartials(_ptr(ws), _ptr(dg), _ptr(db), nparts, Cc, int(acc))
        rc = _fn("lc2is_ln_partials_reduce")(arr, len(chunk), _stream())
        _lib.check(rc, f"ln_partials_reduce n={len(chunk)}")
        items = rest


def ln_defer_end(prev):
    global _ln_defer
    ln_defer_flush()
    _ln_defer = prev


def attention_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, B: int, H: int, Sq: int, Sk: int, D: int,
                  scale: float, *, causal: bool = False, kbias: torch.Tensor | None = None,
                  save_lse: bool = True, out: torch.Tensor | None = None, dropout_p: float = 0.0, seed: int = 0):
    """q [B*Sq, H*D], k/v [B*Sk, H*D] bf16 2-D views (any row stride).  Returns (o [B*Sq, H*D] bf16, lse2).
    dropout_p > 0: dropout on the attention probabilities, decisions = f(seed, (b,h,q), key) (pass the same to attention_bwd)."""
    _chk(q, torch.bfloat16, "q"); _chk(k, torch.bfloat16, "k"); _chk(v, torch.bfloat16, "v")
    _chk(kbias, torch.float32, "kbias")
    if q.shape != (B * Sq, H * D) or k.shape != (B * Sk, H * D) or v.shape != (B * Sk, H * D):
        raise RuntimeError("lc2is_amd.attention_fwd: q/k/v shapes do not match B,H,S,D")
    if kbias is not None and (kbias.shape != (B, Sk) or not kbias.is_contiguous()):
        raise RuntimeError("lc2is_amd.attention_fwd: kbias must be contiguous [B,Sk]")
    o = out if out is not None else torch.empty((B * Sq, H * D), dtype=torch.bfloat16, device=q.device)
    _chk(o, torch.bfloat16, "out")
    lse = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device) if save_lse else None
    if dropout_p > 0.0:
        rc = _fn("lc2is_attention_fwd_dropout")(_ptr(q), _ld(q), _ptr(k), _ld(k), _ptr(v), _ld(v), _ptr(o), _ld(o),
                                                _ptr(lse), _ptr(kbias), B, H, Sq, Sk, D, float(scale), int(causal),
                                                float(dropout_p), int(seed), _stream())
    else:
        rc = _fn("lc2is_attention_fwd")(_ptr(q), _ld(q), _ptr(k), _ld(k), _ptr(v), _ld(v), _ptr(o), _ld(o),
                                        _ptr(lse), _ptr(kbias), B, H, Sq, Sk, D, float(scale), int(causal),
                                        _stream())
    _lib.check(rc, f"attention_fwd B={B} H={H} Sq={Sq} Sk={Sk} D={D}")
    return o, lse


def attention_bwd(q, k, v, o, do, lse2, B: int, H: int, Sq: int, Sk: int, D: int, scale: float, *,
                  causal: bool = False, kbias: torch.Tensor | None = None, dq=None, dk=None, dv=None,
                  dropout_p: float = 0.0, seed: int = 0):
    """Returns (dq, dk, dv) bf16; dq/dk/dv may be preallocated 2-D views (e.g. slices of a packed dQKV)."""
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (o, "o"), (do, "do")):
        _chk(t, torch.bfloat16, n)
    _chk(lse2, torch.float32, "lse2", 3); _chk(kbias, torch.float32, "kbias")
    dev = q.device
    dq = dq if dq is not None else torch.empty((B * Sq, H * D), dtype=torch.bfloat16, device=dev)
    dk = dk if dk is not None else torch.empty((B * Sk, H * D), dtype=torch.bfloat16, device=dev)
    dv = dv if dv is not None else torch.empty((B * Sk, H * D), dtype=torch.bfloat16, device=dev)
    for t, n in ((dq, "dq"), (dk, "dk"), (dv, "dv")):
        _chk(t, torch.bfloat16, n)
    delta = torch.empty((B, H, Sq), dtype=torch.float32, device=dev)
    args = (_ptr(q), _ld(q), _ptr(k), _ld(k), _ptr(v), _ld(v), _ptr(o), _ld(o), _ptr(do), _ld(do), _ptr(dq), _ld(dq),
            _ptr(dk), _ld(dk), _ptr(dv), _ld(dv), _ptr(lse2), _ptr(delta), _ptr(kbias), B, H, Sq, Sk, D, float(scale),
            int(causal))
    if dropout_p > 0.0:
        rc = _fn("lc2is_attention_bwd_dropout")(*args, float(dropout_p), int(seed), _stream())
    else:
        rc = _fn("lc2is_attention_bwd")(*args, _stream())
    _lib.check(rc, f"attention_bwd B={B} H={H} Sq={Sq} Sk={Sk} D={D}")
    return dq, dk, dv


def dropout_rows_f32(x: torch.Tensor, p: float, seed: int, *, resid: torch.Tensor | None = None,
                     out_f32: torch.Tensor | bool | None = True, out_bf16: torch.Tensor | bool | None = None,
                     rows_per_sample: int = 0):
    """y = resid + keep * x / (1 - p) on fp32 rows [M,C]; one decision per element, or per sample of `rows_per_sample` rows
    (drop-path).  Returns (y_f32 or None, y_bf16 or None).  Applied to a gradient with the forward's (p, seed) it is the
    backward of the same site."""
    _chk(x, torch.float32, "x"); _chk(resid, torch.float32, "resid")
    M, Cc = x.shape
    yf = _out(out_f32, (M, Cc), torch.float32, x.device)
    yb = _out(out_bf16, (M, Cc), torch.bfloat16, x.device)
    _chk(yf, torch.float32, "out_f32"); _chk(yb, torch.bfloat16, "out_bf16")
    rc = _fn("lc2is_dropout_rows_f32")(_ptr(x), _ld(x), _ptr(resid), _ld(resid), _ptr(yf), _ld(yf), _ptr(yb), _ld(yb), M, Cc,
                                       int(rows_per_sample), float(p), int(seed), _stream())
    _lib.check(rc, f"dropout_rows_f32 M={M} C={Cc}")
    return yf, yb


def dropout_rows_bf16(x: torch.Tensor, p: float, seed: int, out: torch.Tensor | None = None):
    """bf16 [M,C] -> keep * x / (1 - p), in place unless `out` is given."""
    _chk(x, torch.bfloat16, "x")
    y = x if out is None else out
    _chk(y, torch.bfloat16, "out")
    M, Cc = x.shape
    rc = _fn("lc2is_dropout_rows_bf16")(_ptr(x), _ld(x), _ptr(y), _ld(y), M, Cc, float(p), int(seed), _stream())
    _lib.check(rc, f"dropout_rows_bf16 M={M} C={Cc}")
    return y


def dropout_mask(rows: int, cols: int, p: float, seed: int, device) -> torch.Tensor:
    """The decisions keep(seed, r, c) as a uint8 [rows, cols] tensor — for tests (the product path never stores a mask)."""
    out = torch.empty((rows, cols), dtype=torch.uint8, device=device)
    _lib.check(_fn("lc2is_dropout_mask")(_ptr(out), rows, cols, float(p), int(seed), _stream()), "dropout_mask")
    return out


INTERP_BICUBIC, INTERP_BILINEAR = 0, 1


class ShadowDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("dstT", C.c_void_p), ("N", C.c_int), ("K", C.c_int),
                ("ld_dst", C.c_int), ("ld_dstT", C.c_int), ("tile_start", C.c_int), ("flags", C.c_int)]


class ShadowTable:
    """Device-resident descriptor table for ``lc2is_shadow_refresh`` (built once; pointers must stay valid)."""

    def __init__(self, entries, device):
        # entries: list of (src_f32 [N,K] contiguous, dst_bf16 2-D view or None, dstT_bf16 2-D view or None)
        descs = (ShadowDesc * len(entries))()
        start = 0
        self._keep = []
        for i, (src, dst, dstT) in enumerate(entries):
            if src.dim() == 1:  # vectors are copied as [1,K] rows
                src = src.unsqueeze(0)
                dst = dst.unsqueeze(0) if dst is not None and dst.dim() == 1 else dst
            f32copy = dst is not None and dst.dtype == torch.float32
            _chk(src, torch.float32, "shadow src")
            _chk(dst, torch.float32 if f32copy else torch.bfloat16, "shadow dst")
            _chk(dstT, torch.bfloat16, "shadow dstT")
            if f32copy and dstT is not None:
                raise RuntimeError("lc2is_amd: fp32 shadow copies have no transposed twin")
            if not src.is_contiguous():
                raise RuntimeError("lc2is_amd: shadow source must be contiguous")
            N, K = src.shape
            if K % 4 or (dstT is not None and N % 4):
                raise RuntimeError(f"lc2is_amd: shadow shape [{N},{K}] not supported")
            if dst is not None and tuple(dst.shape) != (N, K):
                raise RuntimeError("lc2is_amd: shadow dst shape mismatch")
            if dstT is not None and tuple(dstT.shape) != (K, N):
                raise RuntimeError("lc2is_amd: shadow dstT shape mismatch")
            descs[i] = ShadowDesc(src.data_ptr(), _ptr(dst), _ptr(dstT), N, K, _ld(dst), _ld(dstT), start, int(f32copy))
            start += ((N + 63) // 64) * ((K + 63) // 64)
            self._keep.append((src, dst, dstT))
        self.n = len(entries)
        self.total_tiles = start
        raw = bytes(descs)
        self.dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)

    def refresh(self):
        rc = _fn("lc2is_shadow_refresh")(self.dev.data_ptr(), self.n, self.total_tiles, _stream())
        _lib.check(rc, "shadow_refresh")


def cast_bf16(src: torch.Tensor, out: torch.Tensor | None = None):
    _chk(src, torch.float32, "src")
    M, Cc = src.shape
    out = out if out is not None else torch.empty((M, Cc), dtype=torch.bfloat16, device=src.device)
    _chk(out, torch.bfloat16, "out")
    _lib.check(_fn("lc2is_cast_f32_bf16")(_ptr(src), _ld(src), _ptr(out), _ld(out), M, Cc, _stream()), "cast")
    return out


def transpose_bf16(src: torch.Tensor, out: torch.Tensor | None = None):
    _chk(src, torch.bfloat16, "src")
    R, Cc = src.shape
    out = out if out is not None else torch.empty((Cc, R), dtype=torch.bfloat16, device=src.device)
    _chk(out, torch.bfloat16, "out")
    _lib.check(_fn("lc2is_transpose_bf16")(_ptr(src), _ld(src), _ptr(out), _ld(out), R, Cc, _stream()),
               "transpose")
    return out


def patchify(pixels: torch.Tensor, patch: int, kpad: int | None = None):
    """pixels fp32 [B,3,H,W] contiguous -> bf16 [B*G*G, kpad]."""
    _chk(pixels, torch.float32, "pixel_values", 4)
    if not pixels.is_contiguous() or pixels.shape[1] != 3:
        raise RuntimeError("lc2is_amd.patchify: pixel_values must be contiguous [B,3,H,W]")
    B, _, H, W = pixels.shape
    G = H // patch
    k = 3 * patch * patch
    kpad = kpad or ((k + 63) // 64) * 64
    out = torch.empty((B * G * G, kpad), dtype=torch.bfloat16, device=pixels.device)
    _lib.check(_fn("lc2is_patchify")(_ptr(pixels), _ptr(out), kpad, B, H, W, patch, _stream()), "patchify")
    return out


def vit_embed_fwd(patch_f32, cls, pos, B: int, P: int):
    _chk(patch_f32, torch.float32, "patch"); _chk(cls, torch.float32, "cls", 1); _chk(pos, torch.float32, "pos")
    Cc = patch_f32.shape[1]
    x = torch.empty((B * (P + 1), Cc), dtype=torch.float32, device=patch_f32.device)
    _lib.check(_fn("lc2is_vit_embed_fwd")(_ptr(patch_f32), _ld(patch_f32), _ptr(cls), _ptr(pos), _ptr(x), Cc, B,
                                          P, Cc, _stream()), "vit_embed_fwd")
    return x


def vit_embed_bwd(dx, dpos, dcls, B: int, P: int, accumulate: bool = False):
    _chk(dx, torch.float32, "dx"); _chk(dpos, torch.float32, "dpos"); _chk(dcls, torch.float32, "dcls", 1)
    Cc = dx.shape[1]
    dpatch = torch.empty((B * P, Cc), dtype=torch.bfloat16, device=dx.device)
    _lib.check(_fn("lc2is_vit_embed_bwd")(_ptr(dx), _ld(dx), _ptr(dpos), _ptr(dcls), _ptr(dpatch), Cc, B, P, Cc,
                                          int(accumulate), _stream()), "vit_embed_bwd")
    return dpatch


def text_embed_fwd(ids, tok, pos):
    _chk(ids, torch.int64, "input_ids"); _chk(tok, torch.float32, "tok"); _chk(pos, torch.float32, "pos")
    if not ids.is_contiguous():
        raise RuntimeError("lc2is_amd.text_embed_fwd: input_ids must be contiguous")
    B, L = ids.shape
    V, Cc = tok.shape
    x = torch.empty((B * L, Cc), dtype=torch.float32, device=tok.device)
    _lib.check(_fn("lc2is_text_embed_fwd")(_ptr(ids), _ptr(tok), _ptr(pos), _ptr(x), Cc, B, L, Cc, V, _stream()),
               "text_embed_fwd")
    return x


def text_embed_bwd(ids, dx, dtok, dpos, accumulate: bool = False):
    """dtok must already hold the running gradient (or zeros); per-token sums in row order, no atomics (reproducible)."""
    _chk(ids, torch.int64, "input_ids"); _chk(dx, torch.float32, "dx")
    B, L = ids.shape
    V, Cc = dtok.shape
    _lib.check(_fn("lc2is_text_embed_bwd")(_ptr(ids), _ptr(dx), _ld(dx), _ptr(dtok), _ptr(dpos), B, L, Cc, V,
                                           int(accumulate), _stream()), "text_embed_bwd")


def rows_copy(src, S_src: int, src_off: int, S_dst: int, dst_off: int, B: int, n: int, *, dst_f32=None,
              dst_bf16=None):
    sb = src.dtype == torch.bfloat16
    _chk(src, torch.bfloat16 if sb else torch.float32, "src")
    Cc = src.shape[1]
    if not src.is_contiguous() or src.shape[0] != B * S_src:
        raise RuntimeError("lc2is_amd.rows_copy: src must be contiguous [B*S_src, C]")
    for t in (dst_f32, dst_bf16):
        if t is not None and (not t.is_contiguous() or tuple(t.shape) != (B * S_dst, Cc)):
            raise RuntimeError("lc2is_amd.rows_copy: dst must be contiguous [B*S_dst, C]")
    _lib.check(_fn("lc2is_rows_copy_bf16" if sb else "lc2is_rows_copy_f32")(_ptr(src), S_src, src_off, _ptr(dst_f32), _ptr(dst_bf16),
                                                                            S_dst, dst_off, B, n, Cc, _stream()), "rows_copy")


def sgd_step(params, grads, momentum_buf, lr, momentum=0.0, weight_decay=0.0, grad_scale=1.0):
    _chk(params, torch.float32, "params", 1); _chk(grads, torch.float32, "grads", 1)
    _lib.check(_fn("lc2is_sgd_step")(_ptr(params), _ptr(grads), _ptr(momentum_buf), params.numel(), lr, momentum,
                                     weight_decay, grad_scale, _stream()), "sgd_step")


def adamw_step(params, grads, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    _chk(params, torch.float32, "params", 1); _chk(grads, torch.float32, "grads", 1)
    _lib.check(_fn("lc2is_adamw_step")(_ptr(params), _ptr(grads), _ptr(m), _ptr(v), params.numel(), lr, beta1,
                                       beta2, eps, weight_decay, int(step), grad_scale, _stream()), "adamw_step")


def head_upsample_ce(scores_lo, labels, B: int, h: int, w: int, C: int, S: int, mode: int = INTERP_BICUBIC, *,
                     want_grad: bool = False, want_scores: bool = False, want_loss: bool = True,
                     ignore_index: int = -100, grad_scale: float = 1.0):
    """scores_lo fp32 [B*h*w, ld].  Returns (loss_sum[2] or None, dscores_lo or None, scores_hi NCHW or None)."""
    _chk(scores_lo, torch.float32, "scores_lo"); _chk(labels, torch.int64, "labels", 3)
    if not scores_lo.is_contiguous():
        raise RuntimeError("lc2is_amd.head_upsample_ce: scores_lo must be contiguous")
    ld = scores_lo.shape[1]
    dev = scores_lo.device
    if labels is not None and (tuple(labels.shape) != (B, h * S, w * S) or not labels.is_contiguous()):
        raise RuntimeError(f"lc2is_amd.head_upsample_ce: labels must be contiguous [{B},{h*S},{w*S}]")
    # S = 4 / 8 / 16: per-block partials in a workspace + a fixed-order second launch (reproducible; outputs overwritten);
    # other S: the atomic path adds into cleared buffers
    nbytes = _fn("lc2is_head_upsample_ce_workspace_bytes")(B, h, w, C, S, mode, int(want_grad)) if want_loss else 0
    alloc = torch.empty if nbytes else torch.zeros
    loss = alloc(2, dtype=torch.float32, device=dev) if want_loss else None
    dlo = alloc(scores_lo.shape, dtype=torch.float32, device=dev) if want_grad else None
    hi = torch.empty((B, C, h * S, w * S), dtype=torch.float32, device=dev) if want_scores else None
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev) if nbytes else None   # (per call: the slabs live until the second launch, in stream order)
    rc = _fn("lc2is_head_upsample_ce")(_ptr(scores_lo), ld, _ptr(labels), _ptr(dlo), _ptr(hi), _ptr(loss), B, h,
                                       w, C, S, mode, ignore_index, grad_scale, _ptr(ws), nbytes, _stream())
    _lib.check(rc, f"head_upsample_ce B={B} h={h} w={w} C={C} S={S}")
    return loss, dlo, hi


def ce_nchw_fwd(logits, labels, ignore_index: int = -100):
    _chk(logits, torch.float32, "logits", 4); _chk(labels, torch.int64, "labels", 3)
    if not logits.is_contiguous() or not labels.is_contiguous():
        raise RuntimeError("lc2is_amd.ce_nchw_fwd: logits/labels must be contiguous")
    B, Cc, H, W = logits.shape
    lse = torch.empty((B, H, W), dtype=torch.float32, device=logits.device)
    loss = torch.zeros(2, dtype=torch.float32, device=logits.device)
    _lib.check(_fn("lc2is_ce_nchw_fwd")(_ptr(logits), _ptr(labels), _ptr(lse), _ptr(loss), B, Cc, H * W,
                                        ignore_index, _stream()), "ce_nchw_fwd")
    return loss, lse


def ce_nchw_bwd(logits, labels, lse, grad_scale_dev, grad_scale: float, ignore_index: int = -100):
    B, Cc, H, W = logits.shape
    d = torch.empty_like(logits)
    _lib.check(_fn("lc2is_ce_nchw_bwd")(_ptr(logits), _ptr(labels), _ptr(lse), _ptr(grad_scale_dev), grad_scale,
                                        _ptr(d), B, Cc, H * W, ignore_index, _stream()), "ce_nchw_bwd")
    return d


def upsample_bwd_nchw(dhi, B: int, h: int, w: int, C: int, S: int, mode: int, ld: int):
    """dhi fp32 NCHW [B,C,h*S,w*S] -> dlo fp32 [B*h*w, ld] (zero padded columns)."""
    _chk(dhi, torch.float32, "dhi", 4)
    if not dhi.is_contiguous() or tuple(dhi.shape) != (B, C, h * S, w * S):
        raise RuntimeError("lc2is_amd.upsample_bwd_nchw: dhi must be contiguous [B,C,h*S,w*S]")
    dlo = torch.zeros((B * h * w, ld), dtype=torch.float32, device=dhi.device)
    _lib.check(_fn("lc2is_upsample_bwd_nchw")(_ptr(dhi), _ptr(dlo), ld, B, h, w, C, S, mode, _stream()),
               "upsample_bwd_nchw")
    return dlo


def _dense(t, dtype, name):
    _chk(t, dtype, name)
    if t is not None and not t.is_contiguous():
        raise RuntimeError(f"lc2is_amd: {name} must be contiguous")


def bilinear_up_fwd(x, B: int, h: int, w: int, S: int, *, want_f32: bool = True, want_bf16: bool = False):
    """x fp32 [B*h*w, C] channels-last tokens -> ([B*h*S*w*S, C] fp32 or None, bf16 twin or None)."""
    _dense(x, torch.float32, "x")
    Cc = x.shape[1]
    n = B * h * S * w * S
    of = torch.empty((n, Cc), dtype=torch.float32, device=x.device) if want_f32 else None
    ob = torch.empty((n, Cc), dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    _lib.check(_fn("lc2is_bilinear_up_fwd")(_ptr(x), _ptr(of), _ptr(ob), B, h, w, Cc, S, _stream()), "bilinear_up_fwd")
    return of, ob


def bilinear_up_bwd(dout, B: int, h: int, w: int, S: int, *, din=None, accumulate: bool = False, want_bf16: bool = False):
    _dense(dout, torch.float32, "dout")
    Cc = dout.shape[1]
    if din is None:
        din = torch.empty((B * h * w, Cc), dtype=torch.float32, device=dout.device)
        accumulate = False
    _dense(din, torch.float32, "din")
    d16 = torch.empty((B * h * w, Cc), dtype=torch.bfloat16, device=dout.device) if want_bf16 else None
    _lib.check(_fn("lc2is_bilinear_up_bwd")(_ptr(dout), _ptr(din), _ptr(d16), B, h, w, Cc, S, int(accumulate), _stream()),
               "bilinear_up_bwd")
    return din, d16


def sr_gather(x16, B: int, h: int, w: int, scatter: bool = False):
    """gather: bf16 [B*h*w, C] -> [B*h*w/4, 4C];  scatter: bf16 [B*h*w/4, 4C] -> [B*h*w, C]."""
    _dense(x16, torch.bfloat16, "x")
    if scatter:
        Cc = x16.shape[1] // 4
        out = torch.empty((B * h * w, Cc), dtype=torch.bfloat16, device=x16.device)
    else:
        Cc = x16.shape[1]
        out = torch.empty((B * h * w // 4, 4 * Cc), dtype=torch.bfloat16, device=x16.device)
    _lib.check(_fn("lc2is_sr_gather")(_ptr(x16), _ptr(out), B, h, w, Cc, int(scatter), _stream()), "sr_gather")
    return out


def l2norm_fwd(x, eps: float = 1e-12, *, want_f32: bool = True, want_bf16: bool = True):
    _dense(x, torch.float32, "x")
    M, Cc = x.shape
    yf = torch.empty_like(x) if want_f32 else None
    yb = torch.empty((M, Cc), dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    inv = torch.empty((M,), dtype=torch.float32, device=x.device)
    _lib.check(_fn("lc2is_l2norm_fwd")(_ptr(x), _ptr(yf), _ptr(yb), _ptr(inv), M, Cc, eps, _stream()), "l2norm_fwd")
    return yf, yb, inv


def l2norm_bwd(dy, x, inv, eps: float = 1e-12):
    _dense(dy, torch.float32, "dy"); _dense(x, torch.float32, "x")
    M, Cc = x.shape
    dx = torch.empty_like(x)
    _lib.check(_fn("lc2is_l2norm_bwd")(_ptr(dy), _ptr(x), _ptr(inv), _ptr(dx), M, Cc, eps, _stream()), "l2norm_bwd")
    return dx


def add_n(tensors, *, want_f32: bool = True, want_bf16: bool = False):
    """Elementwise sum of 2..4 fp32 tensors of equal shape."""
    if not 2 <= len(tensors) <= 4:
        raise RuntimeError("lc2is_amd.add_n takes 2 to 4 tensors")
    for t in tensors:
        _chk(t, torch.float32, "addend", None)
        if not t.is_contiguous() or t.shape != tensors[0].shape:
            raise RuntimeError("lc2is_amd.add_n: addends must be contiguous and equally shaped")
    ps = [_ptr(t) for t in tensors] + [None] * (4 - len(tensors))
    of = torch.empty_like(tensors[0]) if want_f32 else None
    ob = torch.empty(tensors[0].shape, dtype=torch.bfloat16, device=tensors[0].device) if want_bf16 else None
    _lib.check(_fn("lc2is_add_n")(*ps, _ptr(of), _ptr(ob), tensors[0].numel(), _stream()), "add_n")
    return of, ob


def sr_scatter_add(src16, dst32, B: int, h: int, w: int):
    """dst32 [B*h*w, C] fp32 += scatter(src16 [B*h*w/4, 4C] bf16)."""
    _dense(src16, torch.bfloat16, "src"); _dense(dst32, torch.float32, "dst")
    Cc = dst32.shape[1]
    _lib.check(_fn("lc2is_sr_scatter_add_f32")(_ptr(src16), _ptr(dst32), B, h, w, Cc, _stream()), "sr_scatter_add")


def rows_ce(x, labels, *, loss_sum=None, dx=None, grad_scale: float = 1.0, accumulate_dx: bool = False, want_lse=False):
    _dense(x, torch.float32, "x")
    M, K = x.shape
    lse = torch.empty((M,), dtype=torch.float32, device=x.device) if want_lse else None
    _lib.check(_fn("lc2is_rows_ce")(_ptr(x), _ptr(labels), _ptr(lse), _ptr(loss_sum), _ptr(dx), grad_scale, M, K,
                                    int(accumulate_dx), _stream()), "rows_ce")
    return lse


def cols_ce(x, labels, B: int, H: int, W: int, K: int, loss_sum, dx=None, grad_scale: float = 1.0):
    _dense(x, torch.float32, "x")
    _lib.check(_fn("lc2is_cols_ce")(_ptr(x), _ptr(labels), _ptr(loss_sum), _ptr(dx), grad_scale, B, H, W, K, _stream()),
               "cols_ce")


def npair(x, x_pos, x_neg):
    for t, n in ((x, "x"), (x_pos, "x_pos"), (x_neg, "x_neg")):
        _dense(t, torch.float32, n)
    res = torch.empty((x.shape[0],), dtype=torch.float32, device=x.device)
    _lib.check(_fn("lc2is_npair")(_ptr(x), _ptr(x_pos), _ptr(x_neg), _ptr(res), x.shape[0], x_pos.shape[0],
                                  x_neg.shape[0], x.shape[1], _stream()), "npair")
    return res



def npair_bwd(x, x_pos, x_neg, dres):
    """Gradients of npair's res [n] wrt (x, x_pos, x_neg), all fp32."""
    for t, n in ((x, "x"), (x_pos, "x_pos"), (x_neg, "x_neg")):
        _dense(t, torch.float32, n)
    _chk(dres, torch.float32, "dres", 1)
    n, d = x.shape
    dx, dxp, dxn = torch.empty_like(x), torch.empty_like(x_pos), torch.empty_like(x_neg)
    ws = torch.empty(n * (x_pos.shape[0] + 1), dtype=torch.float32, device=x.device)
    _lib.check(_fn("lc2is_npair_bwd")(_ptr(x), _ptr(x_pos), _ptr(x_neg), _ptr(dres.contiguous()), _ptr(dx), _ptr(dxp), _ptr(dxn),
                                      _ptr(ws), n, x_pos.shape[0], x_neg.shape[0], d, _stream()), "npair_bwd")
    return dx, dxp, dxn

def miou_counts(scores_hi, labels_lo, S: int):
    _chk(scores_hi, torch.float32, "scores_hi", 4); _chk(labels_lo, torch.int64, "labels", 3)
    B, K, H, W = scores_hi.shape
    counts = torch.zeros((B, 3, K), dtype=torch.int32, device=scores_hi.device)
    _lib.check(_fn("lc2is_miou_counts")(_ptr(scores_hi.contiguous()), _ptr(labels_lo.contiguous()), _ptr(counts), B, K, H,
                                        W, S, _stream()), "miou_counts")
    return counts


# ---- Swin backbone ----------------------------------------------------------------------------------------------
def rows_gather(src: torch.Tensor, index_map: torch.Tensor, *, out: torch.Tensor | None = None, out_dtype=None,
                add: torch.Tensor | None = None, cols: int | None = None):
    """out[r, :cols] = (map[r] >= 0 ? src[map[r], :cols] : 0) (+ add[r, :cols]).  src/out fp32 or bf16 2-D (row
    strides allowed), map int32 [rows], add fp32."""
    if src.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("lc2is_amd.rows_gather: src must be fp32 or bf16")
    _chk(src, src.dtype, "src"); _chk(index_map, torch.int32, "map", 1); _chk(add, torch.float32, "add")
    rows = index_map.numel()
    cols = src.shape[1] if cols is None else cols
    if out is None:
        out = torch.empty((rows, cols), dtype=out_dtype or src.dtype, device=src.device)
    if out.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("lc2is_amd.rows_gather: out must be fp32 or bf16")
    _chk(out, out.dtype, "out")
    if out.shape[0] != rows or out.shape[1] < cols or src.shape[1] < cols or (add is not None and add.shape[0] != rows):
        raise RuntimeError("lc2is_amd.rows_gather: shape mismatch")
    rc = _fn("lc2is_rows_gather")(_ptr(src), _ld(src), int(src.dtype == torch.bfloat16), _ptr(out), _ld(out),
                                  int(out.dtype == torch.bfloat16), _ptr(index_map), _ptr(add), _ld(add), rows, cols,
                                  _stream())
    _lib.check(rc, f"rows_gather rows={rows} cols={cols}")
    return out


def swin_attn_fwd(qkv: torch.Tensor, bias: torch.Tensor, nwin: int, win_per_img: int, nwx: int, Hp: int, Wp: int,
                  ws: int, shift: int, nH: int, scale: float, *, save_lse: bool = True, out: torch.Tensor | None = None):
    """qkv bf16 [nwin*ws*ws, 3C]; bias fp32 [nH, S, S].  Returns (o bf16 [nwin*S, C], lse [nwin, nH, S])."""
    _chk(qkv, torch.bfloat16, "qkv"); _dense(bias, torch.float32, "bias") if bias.dim() == 2 else _chk(bias, torch.float32, "bias", 3)
    S, Cc = ws * ws, qkv.shape[1] // 3
    if qkv.shape[0] != nwin * S or tuple(bias.shape) != (nH, S, S) or not bias.is_contiguous():
        raise RuntimeError("lc2is_amd.swin_attn_fwd: shape mismatch")
    o = out if out is not None else torch.empty((nwin * S, Cc), dtype=torch.bfloat16, device=qkv.device)
    _chk(o, torch.bfloat16, "out")
    if tuple(o.shape) != (nwin * S, Cc):
        raise RuntimeError("lc2is_amd.swin_attn_fwd: out shape mismatch")
    lse = torch.empty((nwin, nH, S), dtype=torch.float32, device=qkv.device)
    rc = _fn("lc2is_swin_attn_fwd")(_ptr(qkv), _ld(qkv), _ptr(o), _ld(o), _ptr(lse), _ptr(bias), nwin, win_per_img, nwx,
                                    Hp, Wp, ws, shift, nH, Cc, float(scale), _stream())
    _lib.check(rc, f"swin_attn_fwd nwin={nwin} nH={nH} ws={ws}")
    return o, (lse if save_lse else None)


def swin_attn_bwd(qkv, o, do, lse, bias, nwin: int, win_per_img: int, nwx: int, Hp: int, Wp: int, ws: int, shift: int,
                  nH: int, scale: float, *, dbias: torch.Tensor | None = None, accumulate_dbias: bool = False,
                  dqkv: torch.Tensor | None = None):
    """Returns dqkv bf16 [nwin*S, 3C]; dbias fp32 [nH,S,S] is written (or accumulated) when given."""
    for t, n in ((qkv, "qkv"), (o, "o"), (do, "do")):
        _chk(t, torch.bfloat16, n)
    _chk(lse, torch.float32, "lse", 3); _chk(bias, torch.float32, "bias", 3); _chk(dbias, torch.float32, "dbias", 3)
    Cc = qkv.shape[1] // 3
    if dqkv is None:
        dqkv = torch.empty(qkv.shape, dtype=torch.bfloat16, device=qkv.device)
    _chk(dqkv, torch.bfloat16, "dqkv")
    if dqkv.shape != qkv.shape:
        raise RuntimeError("lc2is_amd.swin_attn_bwd: dqkv shape mismatch")
    ws_b = None
    if dbias is not None:
        nbytes = _fn("lc2is_swin_attn_bwd_workspace_bytes")(nwin, ws, nH)
        ws_b = workspace(nbytes, qkv.device, "swin_attn")
    rc = _fn("lc2is_swin_attn_bwd")(_ptr(qkv), _ld(qkv), _ptr(o), _ld(o), _ptr(do), _ld(do), _ptr(lse), _ptr(bias),
                                    _ptr(dqkv), _ld(dqkv), _ptr(dbias), int(accumulate_dbias), nwin, win_per_img, nwx,
                                    Hp, Wp, ws, shift, nH, Cc, float(scale), _ptr(ws_b),
                                    0 if ws_b is None else ws_b.numel(), _stream())
    _lib.check(rc, f"swin_attn_bwd nwin={nwin} nH={nH} ws={ws}")
    return dqkv


# ---- preprocessing (uint8 images) ---------------------------------------------------------------------------------
def swin_bias_table_grad(dbias: torch.Tensor, offsets: torch.Tensor, positions: torch.Tensor, dtable: torch.Tensor,
                         accumulate: bool = False):
    """dtable [T, nH] (+)= sum of dbias [nH, S, S] over the pairs of each relative offset (CSR index, fixed order)."""
    _chk(dbias, torch.float32, "dbias", 3); _chk(dtable, torch.float32, "dtable", 2)
    _chk(offsets, torch.int32, "offsets", 1); _chk(positions, torch.int32, "positions", 1)
    nH, S, S2 = dbias.shape
    T = dtable.shape[0]
    if S2 != S or dtable.shape[1] != nH or offsets.numel() != T + 1 or positions.numel() != S * S \
            or not dbias.is_contiguous() or not dtable.is_contiguous():
        raise RuntimeError("lc2is_amd.swin_bias_table_grad: shape mismatch")
    rc = _fn("lc2is_swin_bias_table_grad")(_ptr(dbias), _ptr(offsets), _ptr(positions), _ptr(dtable), nH, S * S, T,
                                           int(accumulate), _stream())
    _lib.check(rc, "swin_bias_table_grad")
    return dtable


def _u8(t, name):
    if not t.is_cuda or t.dtype != torch.uint8 or t.dim() != 3 or not t.is_contiguous():
        raise RuntimeError(f"lc2is_amd: {name} must be a contiguous uint8 HWC tensor on a HIP device")


def resample_u8(src: torch.Tensor, out_size: int, axis: int, bounds: torch.Tensor, kk: torch.Tensor):
    """One separable 8-bit Pillow resampling pass over an HWC uint8 image (axis 1 = width, 0 = height)."""
    _u8(src, "src"); _chk(bounds, torch.int32, "bounds"); _chk(kk, torch.int32, "kk")
    H, W, Cc = src.shape
    if tuple(bounds.shape) != (out_size, 2) or kk.shape[0] != out_size or not (bounds.is_contiguous() and kk.is_contiguous()):
        raise RuntimeError("lc2is_amd.resample_u8: coefficient tables do not match out_size")
    out = torch.empty((out_size, W, Cc) if axis == 0 else (H, out_size, Cc), dtype=torch.uint8, device=src.device)
    rc = _fn("lc2is_resample_u8")(_ptr(src), H, W, Cc, _ptr(out), out_size, axis, _ptr(bounds), _ptr(kk), kk.shape[1], _stream())
    _lib.check(rc, f"resample_u8 {H}x{W}x{Cc} -> {out_size} axis {axis}")
    return out


def gather2d_u8(src: torch.Tensor, yi: torch.Tensor, xi: torch.Tensor):
    _u8(src, "src"); _chk(yi, torch.int32, "yi", 1); _chk(xi, torch.int32, "xi", 1)
    H, W, Cc = src.shape
    out = torch.empty((yi.numel(), xi.numel(), Cc), dtype=torch.uint8, device=src.device)
    rc = _fn("lc2is_gather2d_u8")(_ptr(src), H, W, Cc, _ptr(out), yi.numel(), xi.numel(), _ptr(yi), _ptr(xi), _stream())
    _lib.check(rc, "gather2d_u8")
    return out


def crop_lut(src: torch.Tensor, top: int, left: int, S: int, *, lut_f32: torch.Tensor | None = None,
             out_f32: torch.Tensor | None = None, lut_i64: torch.Tensor | None = None, out_i64: torch.Tensor | None = None):
    """S x S crop + lookup: float32 CHW pixel values (lut_f32 [C,256]) and / or int64 label ids of channel 0 (lut_i64 [256])."""
    _u8(src, "src")
    H, W, Cc = src.shape
    if lut_f32 is not None:
        _chk(lut_f32, torch.float32, "lut_f32")
        if tuple(lut_f32.shape) != (Cc, 256) or not lut_f32.is_contiguous():
            raise RuntimeError("lc2is_amd.crop_lut: lut_f32 must be [C,256]")
        out_f32 = out_f32 if out_f32 is not None else torch.empty((Cc, S, S), dtype=torch.float32, device=src.device)
        if tuple(out_f32.shape) != (Cc, S, S) or not out_f32.is_contiguous() or out_f32.dtype != torch.float32:
            raise RuntimeError("lc2is_amd.crop_lut: out_f32 must be contiguous float32 [C,S,S]")
    if lut_i64 is not None:
        _chk(lut_i64, torch.int64, "lut_i64", 1)
        out_i64 = out_i64 if out_i64 is not None else torch.empty((S, S), dtype=torch.int64, device=src.device)
        if tuple(out_i64.shape) != (S, S) or not out_i64.is_contiguous() or out_i64.dtype != torch.int64:
            raise RuntimeError("lc2is_amd.crop_lut: out_i64 must be contiguous int64 [S,S]")
    rc = _fn("lc2is_crop_lut")(_ptr(src), H, W, Cc, top, left, S, _ptr(lut_f32), _ptr(out_f32 if lut_f32 is not None else None),
                               _ptr(lut_i64), _ptr(out_i64 if lut_i64 is not None else None), _stream())
    _lib.check(rc, f"crop_lut {H}x{W} crop {S}@({top},{left})")
    return out_f32, out_i64


def set_cu_budget(ncu: int) -> None:
    """Compute units the GEMM tile planners may count on (0 = all 256): include/lc2is_hip.h lc2is_set_cu_budget."""
    _lib.check(_fn("lc2is_set_cu_budget")(int(ncu)), "set_cu_budget")


def get_cu_budget() -> int:
    return int(_fn("lc2is_get_cu_budget")())
