"""Thin tensor-level wrappers over the C ABI (``include/lc2is_hip.h``).

Each function validates dtype / device / layout (raising ``RuntimeError`` like the reference's torch calls
would), takes raw device pointers + leading dimensions from the tensors and launches on the caller's
current HIP stream.  Outputs and workspaces are allocated here with ``torch.empty`` (PyTorch is the
allocator, nothing else).  There is no CPU path: a non-CUDA tensor is an error.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

ACT_NONE, ACT_QUICK_GELU, ACT_RELU, ACT_DQUICK_GELU, ACT_DRELU = 0, 1, 2, 3, 4

_P, _I, _F, _Z = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_ARGTYPES = {
    "lc2is_gemm_nt_bf16": [_P, _I, _P, _I, _P, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "lc2is_gemm_tn_workspace_bytes": [_I, _I, _I],
    "lc2is_gemm_tn_bf16": [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _P, _Z, _P],
    "lc2is_colsum_workspace_bytes": [_I, _I],
    "lc2is_colsum_bf16": [_P, _I, _P, _I, _I, _I, _P, _Z, _P],
    "lc2is_layernorm_fwd": [_P, _I, _P, _P, _P, _I, _P, _I, _P, _P, _I, _I, _F, _P],
    "lc2is_layernorm_bwd_workspace_bytes": [_I, _I],
    "lc2is_layernorm_bwd": [_P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I,
                            _P, _Z, _P],
    "lc2is_attention_fwd": [_P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "lc2is_attention_bwd": [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P,
                            _I, _I, _I, _I, _I, _F, _I, _P],
}
_bound = {}


def _fn(name: str):
    f = _bound.get(name)
    if f is None:
        f = getattr(_lib.load(), name)
        if name in _ARGTYPES:
            f.argtypes = _ARGTYPES[name]
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
    """out = epi(a @ w.T + bias) (+ resid).  a [M,K] bf16, w [N,K] bf16, bias fp32 [N], resid fp32 [M,N].
    ``out_bf16`` / ``out_f32`` / ``aux_out``: True = allocate, tensor = write into it, None/False = skip.
    Returns (out_bf16, out_f32, aux_out) with None for skipped outputs."""
    _chk(a, torch.bfloat16, "a"); _chk(w, torch.bfloat16, "w")
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


def gemm_tn(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor | None = None, accumulate: bool = False):
    """dw[N,K] (fp32) = dy[M,N]^T @ x[M,K]."""
    _chk(dy, torch.bfloat16, "dy"); _chk(x, torch.bfloat16, "x")
    M, N = dy.shape
    M2, K = x.shape
    if M2 != M:
        raise RuntimeError("lc2is_amd.gemm_tn: M mismatch")
    if dw is None:
        dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
        accumulate = False
    _chk(dw, torch.float32, "dw")
    nbytes = _fn("lc2is_gemm_tn_workspace_bytes")(M, N, K)
    ws = workspace(nbytes, dy.device, "gemm_tn")
    rc = _fn("lc2is_gemm_tn_bf16")(_ptr(dy), _ld(dy), _ptr(x), _ld(x), _ptr(dw), _ld(dw), M, N, K,
                                   int(accumulate), _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, f"gemm_tn M={M} N={N} K={K}")
    return dw


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
    """x fp32 [M,C] -> (y_bf16, y_f32, mean, rstd)."""
    _chk(x, torch.float32, "x"); _chk(gamma, torch.float32, "gamma", 1); _chk(beta, torch.float32, "beta", 1)
    M, Cc = x.shape
    dev = x.device
    yb = _out(out_bf16, (M, Cc), torch.bfloat16, dev)
    yf = _out(out_f32, (M, Cc), torch.float32, dev)
    mean = torch.empty((M,), dtype=torch.float32, device=dev) if save_stats else None
    rstd = torch.empty((M,), dtype=torch.float32, device=dev) if save_stats else None
    rc = _fn("lc2is_layernorm_fwd")(_ptr(x), _ld(x), _ptr(gamma), _ptr(beta), _ptr(yb), _ld(yb), _ptr(yf),
                                    _ld(yf), _ptr(mean), _ptr(rstd), M, Cc, float(eps), _stream())
    _lib.check(rc, f"layernorm_fwd M={M} C={Cc}")
    return yb, yf, mean, rstd


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, mean: torch.Tensor,
                  rstd: torch.Tensor, *, dres: torch.Tensor | None = None, dgamma: torch.Tensor | None = None,
                  dbeta: torch.Tensor | None = None, accumulate: bool = False, want_f32: bool = True,
                  want_bf16: bool = True, need_param_grads: bool = True):
    """Returns (dx_f32, dx_bf16, dgamma, dbeta).  dy is bf16 or fp32 [M,C]."""
    M, Cc = x.shape
    dev = x.device
    dyb = dy if dy.dtype == torch.bfloat16 else None
    dyf = dy if dy.dtype == torch.float32 else None
    if dyb is None and dyf is None:
        raise RuntimeError("lc2is_amd.layernorm_bwd: dy must be bf16 or fp32")
    _chk(dyb, torch.bfloat16, "dy"); _chk(dyf, torch.float32, "dy"); _chk(x, torch.float32, "x")
    _chk(dres, torch.float32, "dres")
    dxf = torch.empty((M, Cc), dtype=torch.float32, device=dev) if want_f32 else None
    dxb = torch.empty((M, Cc), dtype=torch.bfloat16, device=dev) if want_bf16 else None
    if need_param_grads:
        if dgamma is None:
            dgamma = torch.empty((Cc,), dtype=torch.float32, device=dev)
            accumulate = False
            if dbeta is None:
                dbeta = torch.empty((Cc,), dtype=torch.float32, device=dev)
    nbytes = _fn("lc2is_layernorm_bwd_workspace_bytes")(M, Cc)
    ws = workspace(nbytes, dev, "ln_bwd")
    rc = _fn("lc2is_layernorm_bwd")(_ptr(dyb), _ld(dyb), _ptr(dyf), _ld(dyf), _ptr(x), _ld(x), _ptr(gamma),
                                    _ptr(mean), _ptr(rstd), _ptr(dres), _ld(dres), _ptr(dxf), _ld(dxf),
                                    _ptr(dxb), _ld(dxb), _ptr(dgamma), _ptr(dbeta), int(accumulate), M, Cc,
                                    _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, f"layernorm_bwd M={M} C={Cc}")
    return dxf, dxb, dgamma, dbeta


def attention_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, B: int, H: int, Sq: int, Sk: int, D: int,
                  scale: float, *, causal: bool = False, kbias: torch.Tensor | None = None,
                  save_lse: bool = True, out: torch.Tensor | None = None):
    """q [B*Sq, H*D], k/v [B*Sk, H*D] bf16 2-D views (any row stride).  Returns (o [B*Sq, H*D] bf16, lse2)."""
    _chk(q, torch.bfloat16, "q"); _chk(k, torch.bfloat16, "k"); _chk(v, torch.bfloat16, "v")
    _chk(kbias, torch.float32, "kbias")
    if q.shape != (B * Sq, H * D) or k.shape != (B * Sk, H * D) or v.shape != (B * Sk, H * D):
        raise RuntimeError("lc2is_amd.attention_fwd: q/k/v shapes do not match B,H,S,D")
    if kbias is not None and (kbias.shape != (B, Sk) or not kbias.is_contiguous()):
        raise RuntimeError("lc2is_amd.attention_fwd: kbias must be contiguous [B,Sk]")
    o = out if out is not None else torch.empty((B * Sq, H * D), dtype=torch.bfloat16, device=q.device)
    _chk(o, torch.bfloat16, "out")
    lse = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device) if save_lse else None
    rc = _fn("lc2is_attention_fwd")(_ptr(q), _ld(q), _ptr(k), _ld(k), _ptr(v), _ld(v), _ptr(o), _ld(o),
                                    _ptr(lse), _ptr(kbias), B, H, Sq, Sk, D, float(scale), int(causal),
                                    _stream())
    _lib.check(rc, f"attention_fwd B={B} H={H} Sq={Sq} Sk={Sk} D={D}")
    return o, lse


def attention_bwd(q, k, v, o, do, lse2, B: int, H: int, Sq: int, Sk: int, D: int, scale: float, *,
                  causal: bool = False, kbias: torch.Tensor | None = None, dq=None, dk=None, dv=None):
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
    rc = _fn("lc2is_attention_bwd")(_ptr(q), _ld(q), _ptr(k), _ld(k), _ptr(v), _ld(v), _ptr(o), _ld(o),
                                    _ptr(do), _ld(do), _ptr(dq), _ld(dq), _ptr(dk), _ld(dk), _ptr(dv), _ld(dv),
                                    _ptr(lse2), _ptr(delta), _ptr(kbias), B, H, Sq, Sk, D, float(scale),
                                    int(causal), _stream())
    _lib.check(rc, f"attention_bwd B={B} H={H} Sq={Sq} Sk={Sk} D={D}")
    return dq, dk, dv
