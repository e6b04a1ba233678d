"""Multi-scale (hierarchical / FTN) decoders on MI355X — drop-ins for ``model/hierarchical.py`` and
``model/decoder.py:36-134`` (BASELINE config 5).

  SRTransformerCrossA / SRTransformerDecoder(d_model, nhead, sr_ratio=1, dim_feedforward=2048, dropout=0.1, ...)
  SRTransformerSelfA(...)                       post-norm layers whose self-attention keys/values are
                                                LayerNorm(Conv2d(d, d, 2, stride=2)(tokens as a grid))
  CrossABlock(layer, depth=1, upsample=2) / SelfABlock / FTNBlock(attention_block, upsample=2)
                                                `depth` applications of ONE shared layer, then bilinear x2
  HierarchicalCrossA(in_dims, depth, dim, nhead=8, dropout=0.1, batch_first=True).forward(visual, textual)
  HierarchicalSelfA(in_dims, depth, dim, nhead, ...).forward(visual)
  FTNDecoder(in_dims, dim, dropout=0.1).forward(visual, textual)        -> [B, P_0, dim]

Parameter names are the reference's (``attention_stage_4.{i}.layers.{d}.self_attn.in_proj_weight`` ...).  Like
``DecoderLayer``, ``bias=False`` (default) reproduces the parameter set the reference gets under torch 2.10 (the
positional ``device=None`` lands on ``bias``: attention / linear / norm1-3 bias-free; ``sr`` and ``norm`` keep theirs).

HIP path: sr_ratio 2, batch_first, relu, head_dim in {64, 96, 128}; dropout in training mode through in-kernel
counter-based RNG (all sites of torch's layers, no stored masks).
The stride-2 conv is a row gather + MFMA GEMM; the x2 upsamples are channels-last single-pass kernels; the whole
pyramid runs under ONE autograd node.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import ops
from .base import (DropSites, HipModule, assign_rng_names, WgradBatch, drop_branch_add, drop_branch_grad16, grad_buf, linear_bwd_params,
                   require_cuda, vec_grad)


def _isqrt(p: int) -> int:
    h = int(round(p ** 0.5))
    if h * h != p:
        raise ValueError(f"lc2is_amd: token count {p} is not a square grid")
    return h


class _PackedAttnParams(nn.Module):
    """nn.MultiheadAttention parameter layout with equal q/k/v dims: in_proj_weight [3C,C] (+bias), out_proj."""

    def __init__(self, c: int, bias: bool):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * c, c))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * c)) if bias else None
        self.out_proj = nn.Linear(c, c, bias=bias)
        nn.init.xavier_uniform_(self.in_proj_weight)


class _SRLayer(nn.Module):
    cross = True

    def __init__(self, d_model: int, nhead: int, sr_ratio: int = 1, dim_feedforward: int = 2048, dropout: float = 0.1,
                 activation=torch.nn.functional.relu, layer_norm_eps: float = 0.00001, batch_first: bool = False,
                 norm_first: bool = False, device=None, dtype=None, *, bias: bool = False) -> None:
        super().__init__()
        if activation not in (torch.nn.functional.relu, "relu"):
            raise NotImplementedError("lc2is_amd SRTransformer*: only relu is implemented on the HIP path")
        if d_model % nhead or (d_model // nhead) not in (64, 96, 128):
            raise NotImplementedError("lc2is_amd SRTransformer*: head_dim must be 64, 96 or 128")
        if sr_ratio != 2 or norm_first:
            raise NotImplementedError("lc2is_amd SRTransformer*: sr_ratio=2, post-norm is the implemented (reference) case")
        self.d_model, self.nhead, self.sr_ratio, self.dim_feedforward = d_model, nhead, sr_ratio, dim_feedforward
        self.dropout_p, self.eps, self.batch_first = float(dropout), layer_norm_eps, batch_first
        self.self_attn = _PackedAttnParams(d_model, bias)
        if self.cross:
            self.multihead_attn = _PackedAttnParams(d_model, bias)
        self.linear1 = nn.Linear(d_model, dim_feedforward, bias=bias)
        self.linear2 = nn.Linear(dim_feedforward, d_model, bias=bias)
        self.norm1 = nn.LayerNorm(d_model, eps=layer_norm_eps, bias=bias)
        self.norm2 = nn.LayerNorm(d_model, eps=layer_norm_eps, bias=bias)
        if self.cross:
            self.norm3 = nn.LayerNorm(d_model, eps=layer_norm_eps, bias=bias)
        self.sr = nn.Conv2d(d_model, d_model, kernel_size=sr_ratio, stride=sr_ratio)
        self.norm = nn.LayerNorm(d_model)


class SRTransformerCrossA(_SRLayer):
    """model/hierarchical.py:201-225 (parameters + hyper-parameters; compute lives in the owning block)."""
    cross = True


class SRTransformerDecoder(_SRLayer):
    """model/decoder.py:113-134 — same layer under its FTN name."""
    cross = True


class SRTransformerSelfA(_SRLayer):
    """model/hierarchical.py:174-199."""
    cross = False


# ---- one SR layer: shadows, forward, backward -------------------------------------------------------------
def _sr_shadows(layer: _SRLayer, device):
    C, F = layer.d_model, layer.dim_feedforward
    bf = dict(dtype=torch.bfloat16, device=device)
    s = dict(w_in=torch.empty(3 * C, C, **bf), w_inT=torch.empty(C, 3 * C, **bf),
             w_so=torch.empty(C, C, **bf), w_soT=torch.empty(C, C, **bf),
             w1=torch.empty(F, C, **bf), w1T=torch.empty(C, F, **bf), w2=torch.empty(C, F, **bf), w2T=torch.empty(F, C, **bf),
             w_sr=torch.empty(C, 4 * C, **bf), w_srT=torch.empty(4 * C, C, **bf))
    e = [(layer.self_attn.in_proj_weight, s["w_in"], s["w_inT"]), (layer.self_attn.out_proj.weight, s["w_so"], s["w_soT"]),
         (layer.linear1.weight, s["w1"], s["w1T"]), (layer.linear2.weight, s["w2"], s["w2T"])]
    if layer.cross:
        s.update(w_cin=torch.empty(3 * C, C, **bf), w_cinT=torch.empty(C, 3 * C, **bf), w_co=torch.empty(C, C, **bf),
                 w_coT=torch.empty(C, C, **bf))
        e += [(layer.multihead_attn.in_proj_weight, s["w_cin"], s["w_cinT"]),
              (layer.multihead_attn.out_proj.weight, s["w_co"], s["w_coT"])]
    return s, e


def _sr_refresh_conv(layer: _SRLayer, s):
    """Conv weight [Co,Ci,2,2] -> GEMM operand [Co, (2i+j)*Ci + ci] matching ops.sr_gather's row order."""
    C = layer.d_model
    w = layer.sr.weight.detach().view(C, C, 4).transpose(1, 2).reshape(C, 4 * C).contiguous()
    ops.cast_bf16(w, s["w_sr"])
    ops.transpose_bf16(s["w_sr"], s["w_srT"])


def _split_bias(b, C):
    return (None, None) if b is None else (b[:C], b[C:])


def _sr_layer_fwd(x32, x16, mem16, layer: _SRLayer, s, B, P, K, save, ds=None):
    """x32/x16: fp32 stream and its bf16 twin [B*P, C]; mem16 [B*K, C] or None.  Returns (y32, y16, saved)."""
    C, H = layer.d_model, layer.nhead
    D = C // H
    scale = D ** -0.5
    hw = _isqrt(P)
    sa = layer.self_attn
    bq, bkv = _split_bias(sa.in_proj_bias, C)
    sv = {}

    def ln(norm, t):
        yb, yf, m, r = ops.layernorm_fwd(t, norm.weight, norm.bias, norm.eps, save_stats=save, out_bf16=True, out_f32=True)
        return yb, yf, (t, m, r)

    q, _, _ = ops.gemm_nt(x16, s["w_in"][:C], bq)
    g = ops.sr_gather(x16, B, hw, hw)
    _, r32, _ = ops.gemm_nt(g, s["w_sr"], layer.sr.bias, out_bf16=None, out_f32=True)
    rn16, _, mr, rr = ops.layernorm_fwd(r32, layer.norm.weight, layer.norm.bias, layer.norm.eps, save_stats=save)
    kv, _, _ = ops.gemm_nt(rn16, s["w_in"][C:], bkv)
    pd = ds.p if ds is not None else 0.0      # training-mode dropout sites (torch Transformer layers: attention probabilities,
    o1, lse1 = ops.attention_fwd(q, kv[:, :C], kv[:, C:], B, H, P, P // 4, D, scale, save_lse=save, dropout_p=pd,   # dropout1-3,
                                 seed=ds.seed("sa_p") if ds else 0)                                               # feed-forward)
    if ds is None:
        _, s1, _ = ops.gemm_nt(o1, s["w_so"], sa.out_proj.bias, resid=x32, out_bf16=None, out_f32=True)
    else:
        _, br, _ = ops.gemm_nt(o1, s["w_so"], sa.out_proj.bias, out_bf16=None, out_f32=True)
        s1 = drop_branch_add(ds, "d1", br, x32)
    h2, x1, sv["ln1"] = ln(layer.norm1, s1)
    if layer.cross:
        ca = layer.multihead_attn
        cq, ckv = _split_bias(ca.in_proj_bias, C)
        q2, _, _ = ops.gemm_nt(h2, s["w_cin"][:C], cq)
        kv2, _, _ = ops.gemm_nt(mem16, s["w_cin"][C:], ckv)
        o2, lse2 = ops.attention_fwd(q2, kv2[:, :C], kv2[:, C:], B, H, P, K, D, scale, save_lse=save, dropout_p=pd,
                                     seed=ds.seed("ca_p") if ds else 0)
        if ds is None:
            _, s2, _ = ops.gemm_nt(o2, s["w_co"], ca.out_proj.bias, resid=x1, out_bf16=None, out_f32=True)
        else:
            _, br, _ = ops.gemm_nt(o2, s["w_co"], ca.out_proj.bias, out_bf16=None, out_f32=True)
            s2 = drop_branch_add(ds, "d2", br, x1)
        h3, x2, sv["ln2"] = ln(layer.norm2, s2)
        last = layer.norm3
    else:
        h3, x2 = h2, x1
        last = layer.norm2
    a, _, _ = ops.gemm_nt(h3, s["w1"], layer.linear1.bias, act=ops.ACT_RELU)
    if ds is None:
        _, s3, _ = ops.gemm_nt(a, s["w2"], layer.linear2.bias, resid=x2, out_bf16=None, out_f32=True)
    else:
        ops.dropout_rows_bf16(a, ds.p, ds.seed("ff"))          # in place: what linear2 consumes
        _, br, _ = ops.gemm_nt(a, s["w2"], layer.linear2.bias, out_bf16=None, out_f32=True)
        s3 = drop_branch_add(ds, "d3", br, x2)
    y16, y32, sv["ln3"] = ln(last, s3)
    if save:
        sv.update(ds=ds, x16=x16, q=q, g=g, r32=r32, mr=mr, rr=rr, rn16=rn16, kv=kv, o1=o1, lse1=lse1, h2=h2, h3=h3, a=a)
        if layer.cross:
            sv.update(q2=q2, kv2=kv2, o2=o2, lse2=lse2)
    return y32, y16, (sv if save else None)


def _packed_param_grads(attn: _PackedAttnParams, dq16, x_q16, dkv16, x_kv16, C):
    """in_proj_weight [3C,C] rows: q from (dq, x_q), k|v from (dkv, x_kv); packed bias likewise."""
    if not attn.in_proj_weight.requires_grad:
        return
    gw, acc = grad_buf(attn.in_proj_weight)
    gb, accb = vec_grad(attn.in_proj_bias)
    ops.gemm_tn(dq16, x_q16, gw[:C], accumulate=acc, db=gb[:C] if gb is not None and accb == acc else None)
    ops.gemm_tn(dkv16, x_kv16, gw[C:], accumulate=acc, db=gb[C:] if gb is not None and accb == acc else None)
    if gb is not None and accb != acc:
        ops.colsum(dq16, gb[:C], accumulate=accb)
        ops.colsum(dkv16, gb[C:], accumulate=accb)


def _sr_layer_bwd(g32, dmem32, mem16, layer, s, sv, B, P, K):
    """The layer's weight gradients are deferred and leave as one grouped launch (base.WgradBatch)."""
    with WgradBatch():
        return _sr_layer_bwd_impl(g32, dmem32, mem16, layer, s, sv, B, P, K)


def _sr_layer_bwd_impl(g32, dmem32, mem16, layer: _SRLayer, s, sv, B, P, K):
    """g32: gradient wrt the layer output (fp32).  Accumulates the text-memory gradient into dmem32.
    Returns the gradient wrt the layer input (fp32)."""
    C, H = layer.d_model, layer.nhead
    D = C // H
    scale = D ** -0.5
    hw = _isqrt(P)
    sa = layer.self_attn

    def ln_bwd(norm, dy, saved):
        t, m, r = saved
        dg, accg = vec_grad(norm.weight)
        db, _ = vec_grad(norm.bias)
        a32, a16, _, _ = ops.layernorm_bwd(dy, t, norm.weight, m, r, dgamma=dg, dbeta=db, accumulate=accg,
                                           need_param_grads=dg is not None)
        return a32, a16

    ds = sv.get("ds")
    pd = ds.p if ds is not None else 0.0
    last = layer.norm3 if layer.cross else layer.norm2
    d32, d16 = ln_bwd(last, g32, sv["ln3"])                                # wrt s3 = x2 + drop3(W2 drop(relu(W1 h3)))
    d16 = drop_branch_grad16(ds, "d3", d32, d16)
    linear_bwd_params(d16, sv["a"], layer.linear2.weight, layer.linear2.bias)
    dz, _, _ = ops.gemm_nt(d16, s["w2T"], None, act=ops.ACT_DRELU, aux_in=sv["a"])
    if ds is not None:   # sv["a"] is the dropped activation: DRELU zeroed the dropped units, 1/(1-p) is left to apply
        ops.dropout_rows_bf16(dz, ds.p, ds.seeds["ff"])
    linear_bwd_params(dz, sv["h3"], layer.linear1.weight, layer.linear1.bias)
    _, d32, _ = ops.gemm_nt(dz, s["w1T"], None, resid=d32, out_bf16=None, out_f32=True)      # wrt x2
    if layer.cross:
        ca = layer.multihead_attn
        d32, d16 = ln_bwd(layer.norm2, d32, sv["ln2"])                     # wrt s2 = x1 + drop2(Wo attn(...))
        d16 = drop_branch_grad16(ds, "d2", d32, d16)
        linear_bwd_params(d16, sv["o2"], ca.out_proj.weight, ca.out_proj.bias)
        do2, _, _ = ops.gemm_nt(d16, s["w_coT"], None)
        q2, kv2 = sv["q2"], sv["kv2"]
        dq2, dkv2 = torch.empty_like(q2), torch.empty_like(kv2)
        ops.attention_bwd(q2, kv2[:, :C], kv2[:, C:], sv["o2"], do2, sv["lse2"], B, H, P, K, D, scale, dq=dq2,
                          dk=dkv2[:, :C], dv=dkv2[:, C:], dropout_p=pd, seed=ds.seeds["ca_p"] if ds else 0)
        _packed_param_grads(ca, dq2, sv["h2"], dkv2, mem16, C)
        ops.gemm_nt(dkv2, s["w_cinT"][:, C:], None, resid=dmem32, out_bf16=None, out_f32=dmem32)
        _, d32, _ = ops.gemm_nt(dq2, s["w_cinT"][:, :C], None, resid=d32, out_bf16=None, out_f32=True)   # wrt x1
    d32, d16 = ln_bwd(layer.norm1, d32, sv["ln1"])                         # wrt s1 = x + drop1(Wo attn(q(x), kv(sr(x))))
    d16 = drop_branch_grad16(ds, "d1", d32, d16)
    linear_bwd_params(d16, sv["o1"], sa.out_proj.weight, sa.out_proj.bias)
    do1, _, _ = ops.gemm_nt(d16, s["w_soT"], None)
    q, kv = sv["q"], sv["kv"]
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    ops.attention_bwd(q, kv[:, :C], kv[:, C:], sv["o1"], do1, sv["lse1"], B, H, P, P // 4, D, scale, dq=dq, dk=dkv[:, :C],
                      dv=dkv[:, C:], dropout_p=pd, seed=ds.seeds["sa_p"] if ds else 0)
    _packed_param_grads(sa, dq, sv["x16"], dkv, sv["rn16"], C)
    drn16, _, _ = ops.gemm_nt(dkv, s["w_inT"][:, C:], None)
    dg, accg = vec_grad(layer.norm.weight)
    db, _ = vec_grad(layer.norm.bias)
    _, dr16, _, _ = ops.layernorm_bwd(drn16, sv["r32"], layer.norm.weight, sv["mr"], sv["rr"], dgamma=dg, dbeta=db,
                                      accumulate=accg, want_f32=False, need_param_grads=dg is not None)
    if layer.sr.weight.requires_grad:      # conv weight grad, permuted back to the [Co,Ci,2,2] parameter order
        gw, acc = grad_buf(layer.sr.weight)
        gb, accb = grad_buf(layer.sr.bias)
        tmp = ops.gemm_tn(dr16, sv["g"])                                   # [C, (2i+j)*C + ci]
        ops.colsum(dr16, gb, accumulate=accb)
        perm = tmp.view(C, 4, C).transpose(1, 2).reshape(C, C, 2, 2)
        gw.add_(perm) if acc else gw.copy_(perm)
    dg16, _, _ = ops.gemm_nt(dr16, s["w_srT"], None)                       # [B*P/4, 4C]
    _, dx32, _ = ops.gemm_nt(dq, s["w_inT"][:, :C], None, resid=d32, out_bf16=None, out_f32=True)
    ops.sr_scatter_add(dg16, dx32, B, hw, hw)
    return dx32


# ---- blocks -------------------------------------------------------------------------------------------------
class _BlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, memory, anchor, blk, save):
        out, saved = blk._fwd_tensors(x, memory, save)
        ctx.blk, ctx.saved = blk, saved
        return out

    @staticmethod
    def backward(ctx, gout):
        dx, dmem = ctx.blk._bwd_tensors(gout.contiguous(), ctx.saved)
        ctx.saved = None
        return dx, dmem, None, None, None


class _SRBlock(HipModule):
    """`depth` applications of one shared SR layer followed by a bilinear x`upsample` of the token grid."""

    def _init_block(self, layer: _SRLayer, depth: int, upsample: int):
        self.depth, self.upsample = depth, upsample
        self._layer = [layer]  # not registered twice

    @property
    def _l(self) -> _SRLayer:
        return self._layer[0]

    def _build_shadows(self, device):
        s, e = _sr_shadows(self._l, device)
        return s, e

    def _post_refresh(self):
        _sr_refresh_conv(self._l, self._sh)

    # internal: operate on the flattened fp32 stream
    def _fwd(self, x32, x16, mem16, B, P, K, save):
        layer = self._l
        if not layer.batch_first:
            raise NotImplementedError("lc2is_amd SR blocks: only batch_first=True is implemented")
        s = self._ensure_ready()
        saved = []
        for it in range(self.depth):     # the SAME layer `depth` times (shared weights), fresh dropout decisions each time
            ds = DropSites.make(self.training, layer.dropout_p, f"{getattr(self, 'rng_name', '')}sr{it}.")
            x32, x16, sv = _sr_layer_fwd(x32, x16, mem16, layer, s, B, P, K, save, ds)
            saved.append(sv)
        hw = _isqrt(P)
        y32, y16 = ops.bilinear_up_fwd(x32, B, hw, hw, self.upsample, want_bf16=True)
        return y32, y16, saved

    def _bwd(self, g32, dmem32, mem16, saved, B, P, K):
        hw = _isqrt(P)
        g32, _ = ops.bilinear_up_bwd(g32, B, hw, hw, self.upsample)
        for sv in reversed(saved):
            g32 = _sr_layer_bwd(g32, dmem32, mem16, self._l, self._sh, sv, B, P, K)
        self._grads_ready()
        return g32

    # standalone use (the reference's Block.forward)
    def _fwd_tensors(self, x, memory, save):
        require_cuda(x, "tgt")
        B, P, C = x.shape
        x32 = x.reshape(B * P, C).float().contiguous()
        x16 = ops.cast_bf16(x32)
        mem16, K = None, 0
        if memory is not None:
            K = memory.shape[1]
            mem16 = ops.cast_bf16(memory.reshape(B * K, C).float().contiguous())
        y32, _, saved = self._fwd(x32, x16, mem16, B, P, K, save)
        return y32.view(B, P * self.upsample ** 2, C), (dict(layers=saved, mem16=mem16, dims=(B, P, K, C)) if save else None)

    def _bwd_tensors(self, gout, saved):
        B, P, K, C = saved["dims"]
        g32 = gout.reshape(-1, C).float().contiguous()
        dmem = torch.zeros(B * K, C, dtype=torch.float32, device=gout.device) if K else None
        dx = self._bwd(g32, dmem, saved["mem16"], saved["layers"], B, P, K)
        return dx.view(B, P, C), (dmem.view(B, K, C) if K else None)

    def _apply_block(self, x, memory):
        anchor = self._l.norm1.weight
        save = torch.is_grad_enabled() and (anchor.requires_grad or x.requires_grad)
        return _BlockFn.apply(x, memory, anchor, self, save)


class CrossABlock(_SRBlock):
    """model/hierarchical.py:153-172."""

    def __init__(self, layer: nn.Module, depth: int = 1, upsample: int = 2) -> None:
        super().__init__()
        self.layers = nn.ModuleList([layer for _ in range(depth)])
        self._init_block(layer, depth, upsample)

    def forward(self, tgt: torch.Tensor, memory: torch.Tensor):
        return self._apply_block(tgt, memory)


class SelfABlock(_SRBlock):
    """model/hierarchical.py:133-151."""

    def __init__(self, layer: nn.Module, depth: int = 1, upsample: int = 2) -> None:
        super().__init__()
        self.layers = nn.ModuleList([layer for _ in range(depth)])
        self._init_block(layer, depth, upsample)

    def forward(self, src: torch.Tensor):
        return self._apply_block(src, None)


class FTNBlock(_SRBlock):
    """model/decoder.py:96-111."""

    def __init__(self, attention_block: nn.Module, upsample: int = 2) -> None:
        super().__init__()
        self.attention_block = attention_block
        self._init_block(attention_block, 1, upsample)

    def forward(self, tgt: torch.Tensor, memory: torch.Tensor):
        return self._apply_block(tgt, memory)


# ---- pyramids -----------------------------------------------------------------------------------------------
class _PyramidFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v0, v3, textual, anchor, mod, save):
        out, saved = mod._fwd(v0, v3, textual, save)
        ctx.mod, ctx.saved = mod, saved
        return out

    @staticmethod
    def backward(ctx, gout):
        dv0, dv3, dtxt = ctx.mod._bwd(gout.contiguous(), ctx.saved)
        ctx.saved = None
        return dv0, dv3, dtxt, None, None, None


class _Pyramid(HipModule):
    uses_text = True

    def _init_linears(self, in_dims, dim):
        self.in_dims, self.dim = list(in_dims), dim
        if any(c % 64 for c in in_dims[1:]) or in_dims[0] % 8 or dim % 64:
            raise NotImplementedError("lc2is_amd pyramid: stage 2-4 widths and dim must be multiples of 64 (Swin-small "
                                      "[96,192,384,768] and Swin-base [128,256,512,1024] are; stage 1 is zero-padded)")
        self.linear_stage_2 = nn.Linear(in_features=in_dims[2], out_features=in_dims[1])
        self.linear_stage_3 = nn.Linear(in_features=in_dims[3], out_features=in_dims[2])
        self.linear2_stage_1 = nn.Linear(in_features=in_dims[0], out_features=dim)
        self.linear2_stage_2 = nn.Linear(in_features=in_dims[1], out_features=dim)
        self.linear2_stage_3 = nn.Linear(in_features=in_dims[2], out_features=dim)
        self.linear2_stage_4 = nn.Linear(in_features=in_dims[3], out_features=dim)

    _LIN = ("linear_stage_2", "linear_stage_3", "linear2_stage_1", "linear2_stage_2", "linear2_stage_3", "linear2_stage_4")

    def _params_for_version(self):
        return [getattr(self, n).weight for n in self._LIN]

    def _build_shadows(self, device):
        s, e = {}, []
        for n in self._LIN:
            w = getattr(self, n).weight
            N, Kk = w.shape
            Kp = (Kk + 63) // 64 * 64          # e.g. Swin-small stage 1 has 96 channels: zero-padded K
            s[n] = torch.zeros(N, Kp, dtype=torch.bfloat16, device=device)
            s[n + "T"] = torch.zeros(Kp, N, dtype=torch.bfloat16, device=device)
            e.append((w, s[n][:, :Kk], s[n + "T"][:Kk]))
        return s, e

    def _blocks(self):
        return list(self.attention_stage_4) + list(self.attention_stage_3) + list(self.attention_stage_2)

    @staticmethod
    def _pad16(x32, Kp):
        """fp32 [M,K] -> bf16 [M,Kp] (zero padded columns when K is not a multiple of 64)."""
        M, Kk = x32.shape
        if Kk == Kp:
            return ops.cast_bf16(x32)
        out = torch.zeros(M, Kp, dtype=torch.bfloat16, device=x32.device)
        ops.cast_bf16(x32, out[:, :Kk])
        return out

    def _lin(self, name, x16, want_f32=True, want_bf16=True):
        lin = getattr(self, name)
        ob, of, _ = ops.gemm_nt(x16, self._sh[name], lin.bias, out_bf16=True if want_bf16 else None,
                                out_f32=True if want_f32 else None)
        return of, ob

    def _lin_bwd(self, name, dy16, x16, *, resid=None, want_f32=True, want_bf16=False):
        """Parameter grads of Linear `name` and the gradient wrt its (padded) input."""
        lin = getattr(self, name)
        N, Kk = lin.weight.shape
        Kp = self._sh[name].shape[1]
        if Kp == Kk:
            linear_bwd_params(dy16, x16, lin.weight, lin.bias)
        else:
            gw, acc = grad_buf(lin.weight)
            gb, accb = grad_buf(lin.bias)
            tmp = ops.gemm_tn(dy16, x16)
            gw.add_(tmp[:, :Kk]) if acc else gw.copy_(tmp[:, :Kk])
            ops.colsum(dy16, gb, accumulate=accb)
        ob, of, _ = ops.gemm_nt(dy16, self._sh[name + "T"], None, resid=resid, out_bf16=True if want_bf16 else None,
                                out_f32=True if want_f32 else None)
        return of, ob

    def _fwd(self, v0, v3, textual, save):
        require_cuda(v0, "visual")
        for blk in self._blocks():
            if not isinstance(blk, _SRBlock):
                raise TypeError("lc2is_amd pyramid: attention stages must be lc2is_amd SR blocks")
        sh = self._ensure_ready()
        B, P0, C0 = v0.shape
        P3, C3 = v3.shape[1], v3.shape[2]
        h3 = _isqrt(P3)
        P2, P1 = 4 * P3, 16 * P3
        if P0 != 64 * P3:
            raise ValueError("lc2is_amd pyramid: stage 1 must have 64x the tokens of stage 4 (4 stages, x2 each)")
        K = 0
        mem16 = None
        if self.uses_text:
            K = textual.shape[1]
            mem16 = ops.cast_bf16(textual.reshape(B * K, self.dim).float().contiguous())
        v3_32 = v3.reshape(B * P3, C3).float().contiguous()
        v3_16 = self._pad16(v3_32, sh["linear2_stage_4"].shape[1])
        v0_16 = self._pad16(v0.reshape(B * P0, C0).float().contiguous(), sh["linear2_stage_1"].shape[1])
        # top-down path (model/hierarchical.py:102-110)
        _, u3_16 = ops.bilinear_up_fwd(v3_32, B, h3, h3, 2, want_f32=False, want_bf16=True)
        t3a_32, t3a_16 = self._lin("linear_stage_3", u3_16)
        _, u2_16 = ops.bilinear_up_fwd(t3a_32, B, 2 * h3, 2 * h3, 2, want_f32=False, want_bf16=True)
        _, t2a_16 = self._lin("linear_stage_2", u2_16, want_f32=False)
        t1_32, _ = self._lin("linear2_stage_1", v0_16, want_bf16=False)
        x4_32, x4_16 = self._lin("linear2_stage_4", v3_16)
        x3_32, x3_16 = self._lin("linear2_stage_3", t3a_16)
        x2_32, x2_16 = self._lin("linear2_stage_2", t2a_16)
        saved_blocks = []

        def run(blocks, x32, x16, P):
            for blk in blocks:
                x32, x16, svb = blk._fwd(x32, x16, mem16, B, P, K, save)
                saved_blocks.append((blk, svb, P))
                P *= 4
            return x32

        y4 = run(self.attention_stage_4, x4_32, x4_16, P3)
        y3 = run(self.attention_stage_3, x3_32, x3_16, P2)
        y2 = run(self.attention_stage_2, x2_32, x2_16, P1)
        out, _ = ops.add_n([t1_32, y2, y3, y4])
        saved = None
        if save:
            saved = dict(dims=(B, P0, P1, P2, P3, C0, C3, K), mem16=mem16, v0_16=v0_16, v3_16=v3_16, u3_16=u3_16,
                         t3a_16=t3a_16, u2_16=u2_16, t2a_16=t2a_16, blocks=saved_blocks)
        return out.view(B, P0, self.dim), saved

    def _bwd(self, gout, saved):
        B, P0, P1, P2, P3, C0, C3, K = saved["dims"]
        h3 = _isqrt(P3)
        dim = self.dim
        g32 = gout.reshape(B * P0, dim).float().contiguous()
        g16 = ops.cast_bf16(g32)
        dmem = torch.zeros(B * K, dim, dtype=torch.float32, device=gout.device) if K else None
        mem16 = saved["mem16"]
        n4, n3 = len(self.attention_stage_4), len(self.attention_stage_3)
        sb = saved["blocks"]

        def run_bwd(entries, g):
            for blk, svb, P in reversed(entries):
                g = blk._bwd(g, dmem, mem16, svb, B, P, K)
            return g

        # stage-1 branch: out = linear2_stage_1(visual[0]) + ...
        dv0, _ = self._lin_bwd("linear2_stage_1", g16, saved["v0_16"])
        # stage-2 branch
        d2 = run_bwd(sb[n4 + n3:], g32)
        dt2a_16 = self._lin_bwd("linear2_stage_2", ops.cast_bf16(d2), saved["t2a_16"], want_f32=False, want_bf16=True)[1]
        du2, _ = self._lin_bwd("linear_stage_2", dt2a_16, saved["u2_16"])
        dt3a, _ = ops.bilinear_up_bwd(du2, B, 2 * h3, 2 * h3, 2)
        # stage-3 branch joins the gradient of t3a
        d3 = run_bwd(sb[n4:n4 + n3], g32)
        dt3a, _ = self._lin_bwd("linear2_stage_3", ops.cast_bf16(d3), saved["t3a_16"], resid=dt3a)
        du3, _ = self._lin_bwd("linear_stage_3", ops.cast_bf16(dt3a), saved["u3_16"])
        dv3, _ = ops.bilinear_up_bwd(du3, B, h3, h3, 2)
        # stage-4 branch joins the gradient of visual[3]
        d4 = run_bwd(sb[:n4], g32)
        dv3, _ = self._lin_bwd("linear2_stage_4", ops.cast_bf16(d4), saved["v3_16"], resid=dv3)
        self._grads_ready()
        dv0 = dv0[:, :C0].contiguous() if dv0.shape[1] != C0 else dv0
        dv3 = dv3[:, :C3].contiguous() if dv3.shape[1] != C3 else dv3
        return dv0.view(B, P0, C0), dv3.view(B, P3, C3), (dmem.view(B, K, dim) if K else None)

    def _run(self, visual, textual):
        anchor = self.linear2_stage_1.weight
        v0, v3 = visual[0], visual[3]   # visual[1], visual[2] are never read by the reference
        save = torch.is_grad_enabled() and (anchor.requires_grad or v0.requires_grad or v3.requires_grad)
        return _PyramidFn.apply(v0, v3, textual, anchor, self, save)


class HierarchicalCrossA(_Pyramid):
    """model/hierarchical.py:71-131."""
    uses_text = True

    def __init__(self, in_dims: list, depth: list, dim: int, nhead: int = 8, dropout: float = 0.1, batch_first: bool = True) -> None:
        super().__init__()
        assert len(in_dims) == 4
        self._init_linears(in_dims, dim)
        mk = lambda d: CrossABlock(SRTransformerCrossA(d_model=dim, nhead=nhead, sr_ratio=2, dropout=dropout,  # noqa: E731
                                                       batch_first=batch_first), depth=d)
        self.attention_stage_2 = nn.ModuleList([mk(depth[0]) for _ in range(1)])
        self.attention_stage_3 = nn.ModuleList([mk(depth[1]) for _ in range(2)])
        self.attention_stage_4 = nn.ModuleList([mk(depth[2]) for _ in range(3)])
        assign_rng_names(self)   # the six blocks log their dropout sites as attention_stage_S.i.srK.* instead of six times srK.*

    def forward(self, visual, textual: torch.Tensor) -> torch.Tensor:
        return self._run(visual, textual)


class HierarchicalSelfA(_Pyramid):
    """model/hierarchical.py:9-69."""
    uses_text = False

    def __init__(self, in_dims: list, depth: list, dim: int, nhead: int, dropout: float = 0.1, batch_first: bool = True) -> None:
        super().__init__()
        assert len(in_dims) == 4
        self._init_linears(in_dims, dim)
        mk = lambda d: SelfABlock(SRTransformerSelfA(d_model=dim, nhead=nhead, sr_ratio=2, dropout=dropout,  # noqa: E731
                                                     batch_first=batch_first), depth=d)
        self.attention_stage_2 = nn.ModuleList([mk(depth[0]) for _ in range(1)])
        self.attention_stage_3 = nn.ModuleList([mk(depth[1]) for _ in range(2)])
        self.attention_stage_4 = nn.ModuleList([mk(depth[2]) for _ in range(3)])
        assign_rng_names(self)   # the six blocks log their dropout sites as attention_stage_S.i.srK.* instead of six times srK.*

    def forward(self, visual) -> torch.Tensor:
        return self._run(visual, None)


class FTNDecoder(_Pyramid):
    """model/decoder.py:36-94 (nhead is hard-coded to 8 there)."""
    uses_text = True

    def __init__(self, in_dims: list, dim: int, dropout: float = 0.1) -> None:
        super().__init__()
        self._init_linears(in_dims, dim)
        mk = lambda: FTNBlock(SRTransformerDecoder(d_model=dim, nhead=8, sr_ratio=2, dropout=dropout, batch_first=True))  # noqa: E731
        self.attention_stage_2 = nn.ModuleList([mk() for _ in range(1)])
        self.attention_stage_3 = nn.ModuleList([mk() for _ in range(2)])
        self.attention_stage_4 = nn.ModuleList([mk() for _ in range(3)])
        assign_rng_names(self)   # the six blocks log their dropout sites as attention_stage_S.i.srK.* instead of six times srK.*

    def forward(self, visual, textual: torch.Tensor):
        return self._run(visual, textual)
