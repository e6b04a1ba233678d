"""Patch↔text cross-attention decoder on MI355X — drop-in for ``model/decoder.py:9-21``.

  DecoderLayer(d_model, d_kv, nhead, dim_feedforward=2048, dropout=0, activation=relu, layer_norm_eps=1e-5,
               batch_first=False, norm_first=False, device=None, dtype=None)
  DecoderBlock(decoder_layer, num_layers, norm=None).forward(tgt, memory, tgt_mask=None, memory_mask=None,
               tgt_key_padding_mask=None, memory_key_padding_mask=None)

Parameter names equal torch's TransformerDecoderLayer / MultiheadAttention (``self_attn.in_proj_weight``,
``multihead_attn.{q,k,v}_proj_weight``, ``multihead_attn.in_proj_bias``, ``linear1.weight`` ...).

Bias drift (drift #1): run under torch 2.10 the reference's positional ``device=None`` lands on
TransformerDecoderLayer's ``bias`` argument, so self_attn / linear1-2 / norm1-3 are created WITHOUT biases and
only the rebuilt ``multihead_attn`` keeps them.  ``bias=False`` (default) reproduces that parameter set;
``bias=True`` gives the torch-1.x set.  Both kinds of checkpoint load.

Supported on the HIP path: batch_first=True, relu, norm_first True/False, memory_key_padding_mask (other masks raise),
dropout in training mode at all six sites of torch's layer (attention probabilities of both attentions, dropout1-3 on the
branch outputs, dropout between activation and linear2) through counter-based in-kernel RNG (no stored masks).
"""
from __future__ import annotations

import copy

import torch
from torch import nn

from .. import ops
from .base import (DropSites, HipModule, WgradBatch, drop_branch_add, drop_branch_grad16, grad_buf, linear_bwd_params,
                   require_cuda, vec_grad)


class _SelfAttnParams(nn.Module):
    def __init__(self, c, bias):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * c, c))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * c)) if bias else None
        self.out_proj = nn.Linear(c, c, bias=bias)
        nn.init.xavier_uniform_(self.in_proj_weight)


class _CrossAttnParams(nn.Module):
    def __init__(self, c, ckv):
        super().__init__()
        self.q_proj_weight = nn.Parameter(torch.empty(c, c))
        self.k_proj_weight = nn.Parameter(torch.empty(c, ckv))
        self.v_proj_weight = nn.Parameter(torch.empty(c, ckv))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * c))
        self.out_proj = nn.Linear(c, c, bias=True)
        for w in (self.q_proj_weight, self.k_proj_weight, self.v_proj_weight):
            nn.init.xavier_uniform_(w)
        nn.init.zeros_(self.out_proj.bias)


class DecoderLayer(nn.Module):
    """Parameter holder + hyper-parameters; the compute lives in DecoderBlock (one shadow table per block)."""

    def __init__(self, d_model: int, d_kv: int, nhead: int, dim_feedforward: int = 2048, dropout: float = 0,
                 activation=torch.nn.functional.relu, layer_norm_eps: float = 0.00001, batch_first: bool = False,
                 norm_first: bool = False, device=None, dtype=None, *, bias: bool = False) -> None:
        super().__init__()
        if activation not in (torch.nn.functional.relu, "relu"):
            raise NotImplementedError("lc2is_amd DecoderLayer: only relu is implemented on the HIP path")
        if d_model % nhead or (d_model // nhead) not in (64, 96, 128):
            raise NotImplementedError("lc2is_amd DecoderLayer: head_dim must be 64, 96 or 128")
        self.d_model, self.d_kv, self.nhead, self.dim_feedforward = d_model, d_kv, nhead, dim_feedforward
        self.dropout_p, self.eps, self.batch_first, self.norm_first = float(dropout), layer_norm_eps, batch_first, norm_first
        self.self_attn = _SelfAttnParams(d_model, bias)
        self.multihead_attn = _CrossAttnParams(d_model, d_kv)
        self.linear1 = nn.Linear(d_model, dim_feedforward, bias=bias)
        self.linear2 = nn.Linear(dim_feedforward, d_model, bias=bias)
        self.norm1 = nn.LayerNorm(d_model, eps=layer_norm_eps, bias=bias)
        self.norm2 = nn.LayerNorm(d_model, eps=layer_norm_eps, bias=bias)
        self.norm3 = nn.LayerNorm(d_model, eps=layer_norm_eps, bias=bias)
        if device is not None:
            self.to(device)
        # d_kv == d_model: torch's MultiheadAttention keeps ONE packed ``multihead_attn.in_proj_weight`` [3C, C] instead of the
        # three q / k / v matrices (torch:nn/modules/activation.py `_qkv_same_embed_dim`), so that is the key a reference
        # checkpoint holds (DenseClip's prompt layers, model/model.py:119).  The parameters stay separate here; the key is
        # split on load and merged on save.
        self._register_load_state_dict_pre_hook(self._split_packed_cross_attn)
        self._register_state_dict_hook(self._merge_packed_cross_attn)

    def _split_packed_cross_attn(self, state_dict, prefix, *args):
        k = prefix + "multihead_attn.in_proj_weight"
        if self.d_kv == self.d_model and k in state_dict:
            w = state_dict.pop(k)
            C = self.d_model
            for i, name in enumerate(("q_proj_weight", "k_proj_weight", "v_proj_weight")):
                state_dict[prefix + "multihead_attn." + name] = w[i * C:(i + 1) * C]

    @staticmethod
    def _merge_packed_cross_attn(module, state_dict, prefix, local_metadata):
        if module.d_kv != module.d_model:
            return
        keys = [prefix + "multihead_attn." + n for n in ("q_proj_weight", "k_proj_weight", "v_proj_weight")]
        if all(k in state_dict for k in keys):
            packed = torch.cat([state_dict.pop(k) for k in keys], dim=0)
            state_dict[prefix + "multihead_attn.in_proj_weight"] = packed


def _layer_shadows(layer: DecoderLayer, device):
    C, Ckv, F = layer.d_model, layer.d_kv, layer.dim_feedforward
    bf = dict(dtype=torch.bfloat16, device=device)
    sa, ca = layer.self_attn, layer.multihead_attn
    s = dict(w_in=torch.empty(3 * C, C, **bf), w_inT=torch.empty(C, 3 * C, **bf),
             w_so=torch.empty(C, C, **bf), w_soT=torch.empty(C, C, **bf),
             w_q=torch.empty(C, C, **bf), w_qT=torch.empty(C, C, **bf),
             w_kv=torch.empty(2 * C, Ckv, **bf), w_kvT=torch.empty(Ckv, 2 * C, **bf),
             w_co=torch.empty(C, C, **bf), w_coT=torch.empty(C, C, **bf),
             w1=torch.empty(F, C, **bf), w1T=torch.empty(C, F, **bf),
             w2=torch.empty(C, F, **bf), w2T=torch.empty(F, C, **bf))
    e = [(sa.in_proj_weight, s["w_in"], s["w_inT"]), (sa.out_proj.weight, s["w_so"], s["w_soT"]),
         (ca.q_proj_weight, s["w_q"], s["w_qT"]),
         (ca.k_proj_weight, s["w_kv"][:C], s["w_kvT"][:, :C]), (ca.v_proj_weight, s["w_kv"][C:], s["w_kvT"][:, C:]),
         (ca.out_proj.weight, s["w_co"], s["w_coT"]),
         (layer.linear1.weight, s["w1"], s["w1T"]), (layer.linear2.weight, s["w2"], s["w2T"])]
    return s, e


def _layer_fwd(x, mem16, layer: DecoderLayer, s, B, Sq, Sk, kbias, save, ds: DropSites | None = None):
    """One decoder layer on the fp32 residual stream x [B*Sq, C]; mem16 bf16 [B*Sk, Ckv].
    torch:nn/modules/transformer.py:1131-1145 (norm_first) / :1147-1156 (post-norm); `ds` = this forward's dropout sites
    (_sa_block / _mha_block / _ff_block, :1158-1199)."""
    C, H = layer.d_model, layer.nhead
    D = C // H
    scale = D ** -0.5
    sa, ca = layer.self_attn, layer.multihead_attn
    bq, bkv = ca.in_proj_bias[:C], ca.in_proj_bias[C:]
    nf = layer.norm_first
    sv = {}

    def ln(i, t, want_f32):
        n = getattr(layer, f"norm{i}")
        yb, yf, m, r = ops.layernorm_fwd(t, n.weight, n.bias, layer.eps, save_stats=save, out_bf16=True,
                                         out_f32=True if want_f32 else None)
        sv[f"ln{i}"] = (t, m, r)
        return yb, yf

    # --- self attention block
    if nf:
        h1, _ = ln(1, x, False)
    else:
        h1 = ops.cast_bf16(x)
    qkv, _, _ = ops.gemm_nt(h1, s["w_in"], sa.in_proj_bias)
    pd = ds.p if ds is not None else 0.0
    o1, lse1 = ops.attention_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, H, Sq, Sq, D, scale, save_lse=save,
                                 dropout_p=pd, seed=ds.seed("sa_p") if ds else 0)
    if ds is None:
        _, x1, _ = ops.gemm_nt(o1, s["w_so"], sa.out_proj.bias, resid=x, out_bf16=None, out_f32=True)
    else:
        _, br, _ = ops.gemm_nt(o1, s["w_so"], sa.out_proj.bias, out_bf16=None, out_f32=True)
        x1 = drop_branch_add(ds, "d1", br, x)
    if not nf:
        h2, x1 = ln(1, x1, True)
    # --- cross attention block
    if nf:
        h2, _ = ln(2, x1, False)
    q, _, _ = ops.gemm_nt(h2, s["w_q"], bq)
    kv, _, _ = ops.gemm_nt(mem16, s["w_kv"], bkv)
    o2, lse2 = ops.attention_fwd(q, kv[:, :C], kv[:, C:], B, H, Sq, Sk, D, scale, kbias=kbias, save_lse=save,
                                 dropout_p=pd, seed=ds.seed("ca_p") if ds else 0)
    if ds is None:
        _, x2, _ = ops.gemm_nt(o2, s["w_co"], ca.out_proj.bias, resid=x1, out_bf16=None, out_f32=True)
    else:
        _, br, _ = ops.gemm_nt(o2, s["w_co"], ca.out_proj.bias, out_bf16=None, out_f32=True)
        x2 = drop_branch_add(ds, "d2", br, x1)
    if not nf:
        h3, x2 = ln(2, x2, True)
    # --- feed forward block
    if nf:
        h3, _ = ln(3, x2, False)
    a, _, _ = ops.gemm_nt(h3, s["w1"], layer.linear1.bias, act=ops.ACT_RELU)
    if ds is None:
        _, x3, _ = ops.gemm_nt(a, s["w2"], layer.linear2.bias, resid=x2, out_bf16=None, out_f32=True)
    else:
        ops.dropout_rows_bf16(a, ds.p, ds.seed("ff"))          # in place: `a` is now dropout(relu(.)), what linear2 consumes
        _, br, _ = ops.gemm_nt(a, s["w2"], layer.linear2.bias, out_bf16=None, out_f32=True)
        x3 = drop_branch_add(ds, "d3", br, x2)
    if not nf:
        _, x3 = ln(3, x3, True)
    if save:
        sv.update(h1=h1, qkv=qkv, o1=o1, lse1=lse1, h2=h2, q=q, kv=kv, o2=o2, lse2=lse2, h3=h3, a=a, ds=ds)
    return x3, (sv if save else None)


def _layer_bwd(g32, g16, dmem32, mem16, layer, s, sv, B, Sq, Sk, kbias):
    """The layer's weight gradients are deferred and leave as one grouped launch (base.WgradBatch)."""
    with WgradBatch():
        return _layer_bwd_impl(g32, g16, dmem32, mem16, layer, s, sv, B, Sq, Sk, kbias)


def _layer_bwd_impl(g32, g16, dmem32, mem16, layer: DecoderLayer, s, sv, B, Sq, Sk, kbias):
    """Backward of _layer_fwd; accumulates the memory gradient into dmem32 (fp32 [B*Sk, Ckv]).
    Returns (g32, g16) wrt the layer input."""
    C, H = layer.d_model, layer.nhead
    D = C // H
    scale = D ** -0.5
    sa, ca = layer.self_attn, layer.multihead_attn
    nf = layer.norm_first

    def ln_bwd(i, dy, dres):
        n = getattr(layer, f"norm{i}")
        t, m, r = sv[f"ln{i}"]
        dg, accg = vec_grad(n.weight)
        db, _ = vec_grad(n.bias)
        a32, a16, _, _ = ops.layernorm_bwd(dy, t, n.weight, m, r, dres=dres, dgamma=dg, dbeta=db, accumulate=accg,
                                           need_param_grads=dg is not None)
        return a32, a16

    ds = sv.get("ds")
    pd = ds.p if ds is not None else 0.0
    # --- feed forward block:  x3 = x2 + drop3(W2 drop(relu(W1 h3)))      (post-norm: x3 = LN3(...))
    if not nf:
        g32, g16 = ln_bwd(3, g32, None)
    gb16 = drop_branch_grad16(ds, "d3", g32, g16)
    linear_bwd_params(gb16, sv["a"], layer.linear2.weight, layer.linear2.bias)
    dz, _, _ = ops.gemm_nt(gb16, s["w2T"], None, act=ops.ACT_DRELU, aux_in=sv["a"])
    if ds is not None:   # sv["a"] is the DROPPED activation, so DRELU already zeroed the dropped units: only 1/(1-p) is left
        ops.dropout_rows_bf16(dz, ds.p, ds.seeds["ff"])
    linear_bwd_params(dz, sv["h3"], layer.linear1.weight, layer.linear1.bias)
    if nf:
        dh3, _, _ = ops.gemm_nt(dz, s["w1T"], None)
        g32, g16 = ln_bwd(3, dh3, g32)
    else:
        _, g32, _ = ops.gemm_nt(dz, s["w1T"], None, resid=g32, out_bf16=None, out_f32=True)
        g32, g16 = ln_bwd(2, g32, None)
    # --- cross attention block: x2 = x1 + drop2(Wo attn(Wq h2, Wk mem, Wv mem))
    gb16 = drop_branch_grad16(ds, "d2", g32, g16)
    linear_bwd_params(gb16, sv["o2"], ca.out_proj.weight, ca.out_proj.bias)
    do2, _, _ = ops.gemm_nt(gb16, s["w_coT"], None)
    q, kv = sv["q"], sv["kv"]
    dq = torch.empty_like(q)
    dkv = torch.empty_like(kv)
    ops.attention_bwd(q, kv[:, :C], kv[:, C:], sv["o2"], do2, sv["lse2"], B, H, Sq, Sk, D, scale, kbias=kbias, dq=dq,
                      dk=dkv[:, :C], dv=dkv[:, C:], dropout_p=pd, seed=ds.seeds["ca_p"] if ds else 0)
    # packed in_proj_bias [3C] = (q, k, v): write the three slices
    gb, accb = vec_grad(ca.in_proj_bias)
    if ca.q_proj_weight.requires_grad:
        gw, acc = grad_buf(ca.q_proj_weight)
        ops.gemm_tn(dq, sv["h2"], gw, accumulate=acc)
        gw, acc = grad_buf(ca.k_proj_weight)
        ops.gemm_tn(dkv[:, :C], mem16, gw, accumulate=acc)
        gw, acc = grad_buf(ca.v_proj_weight)
        ops.gemm_tn(dkv[:, C:], mem16, gw, accumulate=acc)
    if gb is not None:
        ops.colsum(dq, gb[:C], accumulate=accb)
        ops.colsum(dkv, gb[C:], accumulate=accb)
    ops.gemm_nt(dkv, s["w_kvT"], None, resid=dmem32, out_bf16=None, out_f32=dmem32)
    if nf:
        dh2, _, _ = ops.gemm_nt(dq, s["w_qT"], None)
        g32, g16 = ln_bwd(2, dh2, g32)
    else:
        _, g32, _ = ops.gemm_nt(dq, s["w_qT"], None, resid=g32, out_bf16=None, out_f32=True)
        g32, g16 = ln_bwd(1, g32, None)
    # --- self attention block: x1 = x + drop1(Wo attn(Win h1))
    gb16 = drop_branch_grad16(ds, "d1", g32, g16)
    linear_bwd_params(gb16, sv["o1"], sa.out_proj.weight, sa.out_proj.bias)
    do1, _, _ = ops.gemm_nt(gb16, s["w_soT"], None)
    qkv = sv["qkv"]
    dqkv = torch.empty_like(qkv)
    ops.attention_bwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], sv["o1"], do1, sv["lse1"], B, H, Sq, Sq, D, scale,
                      dq=dqkv[:, :C], dk=dqkv[:, C:2 * C], dv=dqkv[:, 2 * C:], dropout_p=pd,
                      seed=ds.seeds["sa_p"] if ds else 0)
    linear_bwd_params(dqkv, sv["h1"], sa.in_proj_weight, sa.in_proj_bias)
    if nf:
        dh1, _, _ = ops.gemm_nt(dqkv, s["w_inT"], None)
        g32, g16 = ln_bwd(1, dh1, g32)
    else:
        _, g32, _ = ops.gemm_nt(dqkv, s["w_inT"], None, resid=g32, out_bf16=None, out_f32=True)
        g16 = ops.cast_bf16(g32)
    return g32, g16


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tgt, memory, anchor, mod, kpm, save):
        out, saved = mod._fwd(tgt, memory, kpm, save)
        ctx.mod, ctx.saved = mod, saved
        return out

    @staticmethod
    def backward(ctx, gout):
        dtgt, dmem = ctx.mod._bwd(gout.contiguous(), ctx.saved)
        ctx.saved = None
        return dtgt, dmem, None, None, None, None


class DecoderBlock(HipModule):
    """Drop-in for model/decoder.py:15-21 (nn.TransformerDecoder: ``num_layers`` deep copies, optional norm)."""

    def __init__(self, decoder_layer: DecoderLayer, num_layers: int, norm=None):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(decoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.norm = norm

    def _build_shadows(self, device):
        per, entries = [], []
        for layer in self.layers:
            s, e = _layer_shadows(layer, device)
            per.append(s)
            entries += e
        return dict(layers=per), entries

    def _fwd(self, tgt, memory, kpm, save):
        require_cuda(tgt, "tgt")
        l0 = self.layers[0]
        if not l0.batch_first:
            raise NotImplementedError("lc2is_amd DecoderBlock: only batch_first=True is implemented (the reference's use)")
        sh = self._ensure_ready()
        B, Sq, C = tgt.shape
        Sk, Ckv = memory.shape[1], memory.shape[2]
        x = tgt.reshape(B * Sq, C).float().contiguous()
        mem16 = ops.cast_bf16(memory.reshape(B * Sk, Ckv).float().contiguous())
        kbias = None
        if kpm is not None:
            kbias = torch.zeros(B, Sk, dtype=torch.float32, device=tgt.device)
            kbias.masked_fill_(kpm, float("-inf"))
        saved = []
        for li, (layer, s) in enumerate(zip(self.layers, sh["layers"])):
            ds = DropSites.make(self.training, layer.dropout_p, f"{getattr(self, 'rng_name', '')}layers.{li}.")
            x, sv = _layer_fwd(x, mem16, layer, s, B, Sq, Sk, kbias, save, ds)
            saved.append(sv)
        fin = None
        if self.norm is not None:
            xin = x
            _, x, m, r = ops.layernorm_fwd(xin, self.norm.weight, self.norm.bias, self.norm.eps, save_stats=save,
                                           out_bf16=None, out_f32=True)
            fin = (xin, m, r)
        return x.view(B, Sq, C), (dict(layers=saved, mem16=mem16, kbias=kbias, dims=(B, Sq, Sk, C, Ckv), fin=fin)
                                  if save else None)

    def _bwd(self, gout, saved):
        sh = self._sh
        B, Sq, Sk, C, Ckv = saved["dims"]
        g32 = gout.reshape(B * Sq, C).float().contiguous()
        if saved["fin"] is not None:
            xin, m, r = saved["fin"]
            dg, accg = vec_grad(self.norm.weight)
            db, _ = vec_grad(self.norm.bias)
            g32, g16, _, _ = ops.layernorm_bwd(g32, xin, self.norm.weight, m, r, dgamma=dg, dbeta=db, accumulate=accg,
                                               need_param_grads=dg is not None)
        else:
            g16 = ops.cast_bf16(g32)
        dmem = torch.zeros(B * Sk, Ckv, dtype=torch.float32, device=gout.device)
        for layer, s, sv in zip(reversed(self.layers), reversed(sh["layers"]), reversed(saved["layers"])):
            g32, g16 = _layer_bwd(g32, g16, dmem, saved["mem16"], layer, s, sv, B, Sq, Sk, saved["kbias"])
        self._grads_ready()
        return g32.view(B, Sq, C), dmem.view(B, Sk, Ckv)

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None):
        if tgt_mask is not None or memory_mask is not None or tgt_key_padding_mask is not None:
            raise NotImplementedError("lc2is_amd DecoderBlock: only memory_key_padding_mask is implemented "
                                      "(the one mask the reference passes, model/model.py:38)")
        anchor = self.layers[0].norm1.weight
        save = torch.is_grad_enabled() and (anchor.requires_grad or tgt.requires_grad or memory.requires_grad)
        return _DecoderFn.apply(tgt, memory, anchor, self, memory_key_padding_mask, save)


class PromptLayer(DecoderLayer):
    """model/decoder.py:24-28: the same layer class with the reference's default dropout of 0.1."""

    def __init__(self, d_model: int, d_kv: int, nhead: int, dim_feedforward: int = 2048, dropout: float = 0.1,
                 activation=torch.nn.functional.relu, layer_norm_eps: float = 0.00001, batch_first: bool = False,
                 norm_first: bool = False, device=None, dtype=None, *, bias: bool = False) -> None:
        super().__init__(d_model, d_kv, nhead, dim_feedforward, dropout, activation, layer_norm_eps, batch_first, norm_first,
                         device, dtype, bias=bias)


class PromptDecoder(DecoderBlock):
    """model/decoder.py:30-33 (nn.TransformerDecoder with its default forward signature)."""
