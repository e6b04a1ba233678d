"""The older FTN pyramid on MI355X — drop-ins for ``model/ftn.py:67-157``.

  Transformer(repeat, sr_ratio, dim, upsample=True, nhead=8).forward(x [B,P,dim], h) -> [B, P*4**repeat, dim]
      `repeat` standard post-norm ``nn.TransformerDecoderLayer``s (relu, biases, batch_first) whose cross-attention
      memory is LayerNorm(Conv2d(dim, dim, sr, stride=sr)(x as an h x h grid)) of the block INPUT, each followed
      by a bilinear x2 when `upsample`.
  Decoder().forward(x: list of 4 stage tensors) -> [B, 16384, 512]
      per-stage Linear, + bilinear x2 of the next stage's RAW input for stages 1 and 2, Linear -> 512, the
      Transformer stacks of stages 1-3, 4-way sum.  Grids are hard-coded [128, 64, 32, 16] (model/ftn.py:104).

Faithful to two quirks of the reference (they are behaviour, not bugs to fix here):
  * inside ``Transformer.forward`` the grid height passed to every rearrange stays `h` (model/ftn.py:151-156), so
    from the second layer on the 4x longer sequence is upsampled as an h x (P/h) image, not a square one;
  * ``Decoder.attentions[0]`` (sr_ratio 1, one 512-wide head) is constructed but never called (model/ftn.py:121-122):
    its parameters exist under the reference's names and receive no gradient.

Parameter names follow the reference (``attentions.{i}.trans.{r}.layers.0.self_attn.in_proj_weight`` ...,
``attentions.{i}.sr.*``, ``attentions.{i}.norm.*``, ``linears.{i}.*``, ``linears2.{i}.*``).
HIP path: dropout 0 or eval mode (the reference hard-codes torch's default 0.1; `dropout=` is an extension kwarg so a
training config can switch it off), head_dim in {64, 96, 128}.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import ops
from .base import (DropSites, HipModule, assign_rng_names, WgradBatch, drop_branch_add, drop_branch_grad16, grad_buf, linear_bwd_params,
                   require_cuda, vec_grad)
from .hier import _PackedAttnParams, _packed_param_grads, _split_bias


class _StdLayerParams(nn.Module):
    """Parameter set of nn.TransformerDecoderLayer(d_model, nhead, batch_first=True) (torch defaults: ff 2048, relu,
    post-norm, biases everywhere)."""

    def __init__(self, d_model: int, nhead: int, dim_feedforward: int = 2048, eps: float = 1e-5):
        super().__init__()
        self.d_model, self.nhead, self.dim_feedforward, self.eps = d_model, nhead, dim_feedforward, eps
        self.self_attn = _PackedAttnParams(d_model, True)
        self.multihead_attn = _PackedAttnParams(d_model, True)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model, eps=eps)
        self.norm2 = nn.LayerNorm(d_model, eps=eps)
        self.norm3 = nn.LayerNorm(d_model, eps=eps)


class _StackOfOne(nn.Module):
    """nn.TransformerDecoder(layer, num_layers=1): keys ``layers.0.*``."""

    def __init__(self, layer: _StdLayerParams):
        super().__init__()
        self.layers = nn.ModuleList([layer])


def _layer_shadow_entries(layer: _StdLayerParams, s: dict, tag: str, device):
    C, F = layer.d_model, layer.dim_feedforward
    bf = dict(dtype=torch.bfloat16, device=device)
    names = dict(w_in=(3 * C, C), w_so=(C, C), w_cin=(3 * C, C), w_co=(C, C), w1=(F, C), w2=(C, F))
    for n, (a, b) in names.items():
        s[tag + n] = torch.empty(a, b, **bf)
        s[tag + n + "T"] = torch.empty(b, a, **bf)
    src = dict(w_in=layer.self_attn.in_proj_weight, w_so=layer.self_attn.out_proj.weight,
               w_cin=layer.multihead_attn.in_proj_weight, w_co=layer.multihead_attn.out_proj.weight,
               w1=layer.linear1.weight, w2=layer.linear2.weight)
    return [(src[n], s[tag + n], s[tag + n + "T"]) for n in names]


def _std_layer_fwd(x32, x16, mem16, layer: _StdLayerParams, s, tag, B, P, K, save, ds=None):
    """Post-norm decoder layer (torch:nn/modules/transformer.py TransformerDecoderLayer.forward, norm_first=False)."""
    C, H = layer.d_model, layer.nhead
    D = C // H
    scale = D ** -0.5
    sa, ca = layer.self_attn, layer.multihead_attn
    sv = {}

    def ln(norm, t):
        yb, yf, m, r = ops.layernorm_fwd(t, norm.weight, norm.bias, norm.eps, save_stats=save, out_bf16=True, out_f32=True)
        return yb, yf, (t, m, r)

    qkv, _, _ = ops.gemm_nt(x16, s[tag + "w_in"], sa.in_proj_bias)
    pd = ds.p if ds is not None else 0.0      # ds: this forward's dropout sites (torch's default 0.1 at model/ftn.py:135)
    o1, lse1 = ops.attention_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, H, P, P, D, scale, save_lse=save,
                                 dropout_p=pd, seed=ds.seed("sa_p") if ds else 0)
    if ds is None:
        _, s1, _ = ops.gemm_nt(o1, s[tag + "w_so"], sa.out_proj.bias, resid=x32, out_bf16=None, out_f32=True)
    else:
        _, br, _ = ops.gemm_nt(o1, s[tag + "w_so"], sa.out_proj.bias, out_bf16=None, out_f32=True)
        s1 = drop_branch_add(ds, "d1", br, x32)
    h2, x1, sv["ln1"] = ln(layer.norm1, s1)
    cq, ckv = _split_bias(ca.in_proj_bias, C)
    q2, _, _ = ops.gemm_nt(h2, s[tag + "w_cin"][:C], cq)
    kv2, _, _ = ops.gemm_nt(mem16, s[tag + "w_cin"][C:], ckv)
    o2, lse2 = ops.attention_fwd(q2, kv2[:, :C], kv2[:, C:], B, H, P, K, D, scale, save_lse=save, dropout_p=pd,
                                 seed=ds.seed("ca_p") if ds else 0)
    if ds is None:
        _, s2, _ = ops.gemm_nt(o2, s[tag + "w_co"], ca.out_proj.bias, resid=x1, out_bf16=None, out_f32=True)
    else:
        _, br, _ = ops.gemm_nt(o2, s[tag + "w_co"], ca.out_proj.bias, out_bf16=None, out_f32=True)
        s2 = drop_branch_add(ds, "d2", br, x1)
    h3, x2, sv["ln2"] = ln(layer.norm2, s2)
    a, _, _ = ops.gemm_nt(h3, s[tag + "w1"], layer.linear1.bias, act=ops.ACT_RELU)
    if ds is None:
        _, s3, _ = ops.gemm_nt(a, s[tag + "w2"], layer.linear2.bias, resid=x2, out_bf16=None, out_f32=True)
    else:
        ops.dropout_rows_bf16(a, ds.p, ds.seed("ff"))          # in place: what linear2 consumes
        _, br, _ = ops.gemm_nt(a, s[tag + "w2"], layer.linear2.bias, out_bf16=None, out_f32=True)
        s3 = drop_branch_add(ds, "d3", br, x2)
    y16, y32, sv["ln3"] = ln(layer.norm3, s3)
    if save:
        sv.update(ds=ds, x16=x16, qkv=qkv, o1=o1, lse1=lse1, h2=h2, q2=q2, kv2=kv2, o2=o2, lse2=lse2, h3=h3, a=a)
    return y32, y16, (sv if save else None)


def _std_layer_bwd(g32, dmem32, mem16, layer, s, tag, sv, B, P, K):
    """The layer's weight gradients are deferred and leave as one grouped launch (base.WgradBatch)."""
    with WgradBatch():
        return _std_layer_bwd_impl(g32, dmem32, mem16, layer, s, tag, sv, B, P, K)


def _std_layer_bwd_impl(g32, dmem32, mem16, layer: _StdLayerParams, s, tag, sv, B, P, K):
    """g32: gradient wrt the layer output; accumulates the memory gradient into dmem32; returns d/d(layer input)."""
    C, H = layer.d_model, layer.nhead
    D = C // H
    scale = D ** -0.5
    sa, ca = layer.self_attn, layer.multihead_attn

    def ln_bwd(norm, dy, saved):
        t, m, r = saved
        dg, accg = vec_grad(norm.weight)
        db, _ = vec_grad(norm.bias)
        a32, a16, _, _ = ops.layernorm_bwd(dy, t, norm.weight, m, r, dgamma=dg, dbeta=db, accumulate=accg,
                                           need_param_grads=dg is not None)
        return a32, a16

    ds = sv.get("ds")
    pd = ds.p if ds is not None else 0.0
    d32, d16 = ln_bwd(layer.norm3, g32, sv["ln3"])
    d16 = drop_branch_grad16(ds, "d3", d32, d16)
    linear_bwd_params(d16, sv["a"], layer.linear2.weight, layer.linear2.bias)
    dz, _, _ = ops.gemm_nt(d16, s[tag + "w2T"], None, act=ops.ACT_DRELU, aux_in=sv["a"])
    if ds is not None:   # sv["a"] is the dropped activation: DRELU zeroed the dropped units, 1/(1-p) is left to apply
        ops.dropout_rows_bf16(dz, ds.p, ds.seeds["ff"])
    linear_bwd_params(dz, sv["h3"], layer.linear1.weight, layer.linear1.bias)
    _, d32, _ = ops.gemm_nt(dz, s[tag + "w1T"], None, resid=d32, out_bf16=None, out_f32=True)
    d32, d16 = ln_bwd(layer.norm2, d32, sv["ln2"])
    d16 = drop_branch_grad16(ds, "d2", d32, d16)
    linear_bwd_params(d16, sv["o2"], ca.out_proj.weight, ca.out_proj.bias)
    do2, _, _ = ops.gemm_nt(d16, s[tag + "w_coT"], None)
    q2, kv2 = sv["q2"], sv["kv2"]
    dq2, dkv2 = torch.empty_like(q2), torch.empty_like(kv2)
    ops.attention_bwd(q2, kv2[:, :C], kv2[:, C:], sv["o2"], do2, sv["lse2"], B, H, P, K, D, scale, dq=dq2,
                      dk=dkv2[:, :C], dv=dkv2[:, C:], dropout_p=pd, seed=ds.seeds["ca_p"] if ds else 0)
    _packed_param_grads(ca, dq2, sv["h2"], dkv2, mem16, C)
    ops.gemm_nt(dkv2, s[tag + "w_cinT"][:, C:], None, resid=dmem32, out_bf16=None, out_f32=dmem32)
    _, d32, _ = ops.gemm_nt(dq2, s[tag + "w_cinT"][:, :C], None, resid=d32, out_bf16=None, out_f32=True)
    d32, d16 = ln_bwd(layer.norm1, d32, sv["ln1"])
    d16 = drop_branch_grad16(ds, "d1", d32, d16)
    linear_bwd_params(d16, sv["o1"], sa.out_proj.weight, sa.out_proj.bias)
    do1, _, _ = ops.gemm_nt(d16, s[tag + "w_soT"], None)
    qkv = sv["qkv"]
    dqkv = torch.empty_like(qkv)
    ops.attention_bwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], sv["o1"], do1, sv["lse1"], B, H, P, P, D, scale,
                      dq=dqkv[:, :C], dk=dqkv[:, C:2 * C], dv=dqkv[:, 2 * C:], dropout_p=pd,
                      seed=ds.seeds["sa_p"] if ds else 0)
    linear_bwd_params(dqkv, sv["x16"], sa.in_proj_weight, sa.in_proj_bias)
    _, dx32, _ = ops.gemm_nt(dqkv, s[tag + "w_inT"], None, resid=d32, out_bf16=None, out_f32=True)
    return dx32


class _TransformerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, mod, h, save):
        out, saved = mod._fwd_tensors(x, h, save)
        ctx.mod, ctx.saved = mod, saved
        return out

    @staticmethod
    def backward(ctx, gout):
        dx = ctx.mod._bwd_tensors(gout.contiguous(), ctx.saved)
        ctx.saved = None
        return dx, None, None, None, None


class Transformer(HipModule):
    """model/ftn.py:131-157."""

    def __init__(self, repeat, sr_ratio, dim, upsample=True, nhead=8, *, dropout: float = 0.1) -> None:
        super().__init__()
        if dim % nhead:
            raise ValueError("lc2is_amd ftn.Transformer: dim must be divisible by nhead")
        if sr_ratio not in (1, 2):
            raise NotImplementedError("lc2is_amd ftn.Transformer: sr_ratio 1 or 2 (the reference's values)")
        self.trans = nn.ModuleList([_StackOfOne(_StdLayerParams(dim, nhead)) for _ in range(repeat)])
        self.upsample, self.sr_ratio, self.dim, self.nhead, self.repeat = upsample, sr_ratio, dim, nhead, repeat
        self.dropout_p = float(dropout)
        self.sr = nn.Conv2d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)
        self.norm = nn.LayerNorm(512)   # sic: the reference normalises over 512 whatever `dim` is (model/ftn.py:141)

    def _layer(self, r) -> _StdLayerParams:
        return self.trans[r].layers[0]

    def _build_shadows(self, device):
        s, e = {}, []
        for r in range(self.repeat):
            e += _layer_shadow_entries(self._layer(r), s, f"l{r}.", device)
        if self.sr_ratio == 2:
            C = self.dim
            s["w_sr"] = torch.empty(C, 4 * C, dtype=torch.bfloat16, device=device)
            s["w_srT"] = torch.empty(4 * C, C, dtype=torch.bfloat16, device=device)
        return s, e

    def _post_refresh(self):
        if self.sr_ratio == 2:   # conv weight [Co,Ci,2,2] -> GEMM operand [Co, (2i+j)*Ci + ci] (ops.sr_gather's row order)
            C = self.dim
            w = self.sr.weight.detach().view(C, C, 4).transpose(1, 2).reshape(C, 4 * C).contiguous()
            ops.cast_bf16(w, self._sh["w_sr"])
            ops.transpose_bf16(self._sh["w_sr"], self._sh["w_srT"])

    def _check(self, P, h):
        D = self.dim // self.nhead
        if D not in (64, 96, 128):
            raise NotImplementedError(f"lc2is_amd ftn.Transformer: head_dim {D} has no HIP attention kernel (64/96/128)")
        if self.dim != 512 and self.sr_ratio == 2:
            raise ValueError("lc2is_amd ftn.Transformer: `norm` is LayerNorm(512) in the reference, so dim must be 512")
        if P % h or (self.sr_ratio == 2 and (h % 2 or P != h * h)):
            raise ValueError(f"lc2is_amd ftn.Transformer: {P} tokens are not an h={h} grid")

    # internal: flattened fp32 stream + bf16 twin
    def _fwd(self, x32, x16, B, P, h, save):
        self._check(P, h)
        s = self._ensure_ready()
        C = self.dim
        sv = dict(layers=[])
        if self.sr_ratio == 2:
            g = ops.sr_gather(x16, B, h, h)
            _, r32, _ = ops.gemm_nt(g, s["w_sr"], self.sr.bias, out_bf16=None, out_f32=True)
            mem16, _, mr, rr = ops.layernorm_fwd(r32, self.norm.weight, self.norm.bias, self.norm.eps, save_stats=save)
            K = P // 4
            if save:
                sv.update(g=g, r32=r32, mr=mr, rr=rr)
        else:
            mem16, K = x16, P
        cur = P
        for r in range(self.repeat):
            ds = DropSites.make(self.training, self.dropout_p, f"{getattr(self, 'rng_name', '')}trans.{r}.")
            x32, x16, svl = _std_layer_fwd(x32, x16, mem16, self._layer(r), s, f"l{r}.", B, cur, K, save, ds)
            sv["layers"].append((svl, cur))
            if self.upsample:   # the grid height stays h (model/ftn.py:151-156)
                x32, x16 = ops.bilinear_up_fwd(x32, B, h, cur // h, 2, want_bf16=True)
                cur *= 4
        sv.update(mem16=mem16, K=K, P=P, h=h)
        return x32, x16, (sv if save else None), cur

    def _bwd(self, g32, sv, B):
        s = self._sh
        C, P, h, K, mem16 = self.dim, sv["P"], sv["h"], sv["K"], sv["mem16"]
        dmem = torch.zeros(B * K, C, dtype=torch.float32, device=g32.device)
        for r in reversed(range(self.repeat)):
            svl, cur = sv["layers"][r]
            if self.upsample:
                g32, _ = ops.bilinear_up_bwd(g32, B, h, cur // h, 2)
            g32 = _std_layer_bwd(g32, dmem, mem16, self._layer(r), s, f"l{r}.", svl, B, cur, K)
        if self.sr_ratio == 2:
            dg, accg = vec_grad(self.norm.weight)
            db, _ = vec_grad(self.norm.bias)
            _, dr16, _, _ = ops.layernorm_bwd(dmem, sv["r32"], self.norm.weight, sv["mr"], sv["rr"], dgamma=dg, dbeta=db,
                                              accumulate=accg, want_f32=False, need_param_grads=dg is not None)
            if self.sr.weight.requires_grad:
                gw, acc = grad_buf(self.sr.weight)
                gb, accb = grad_buf(self.sr.bias)
                tmp = ops.gemm_tn(dr16, sv["g"])
                ops.colsum(dr16, gb, accumulate=accb)
                perm = tmp.view(C, 4, C).transpose(1, 2).reshape(C, C, 2, 2)
                gw.add_(perm) if acc else gw.copy_(perm)
            dg16, _, _ = ops.gemm_nt(dr16, s["w_srT"], None)
            ops.sr_scatter_add(dg16, g32, B, h, h)
        else:
            g32 = g32 + dmem
        self._grads_ready()
        return g32

    def _fwd_tensors(self, x, h, save):
        require_cuda(x, "x")
        B, P, C = x.shape
        x32 = x.reshape(B * P, C).float().contiguous()
        y32, _, saved, cur = self._fwd(x32, ops.cast_bf16(x32), B, P, int(h), save)
        return y32.view(B, cur, C), ((saved, B, P, C) if save else None)

    def _bwd_tensors(self, gout, saved):
        sv, B, P, C = saved
        return self._bwd(gout.reshape(-1, C).float().contiguous(), sv, B).view(B, P, C)

    def forward(self, x, h):
        anchor = self.norm.weight
        save = torch.is_grad_enabled() and (anchor.requires_grad or x.requires_grad)
        return _TransformerFn.apply(x, anchor, self, int(h), save)


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, x1, x2, x3, anchor, mod, save):
        out, saved = mod._fwd([x0, x1, x2, x3], save)
        ctx.mod, ctx.saved = mod, saved
        return out

    @staticmethod
    def backward(ctx, gout):
        dxs = ctx.mod._bwd(gout.contiguous(), ctx.saved)
        ctx.saved = None
        return (*dxs, None, None, None)


class Decoder(HipModule):
    """model/ftn.py:67-129 (Swin-base stage widths [128, 256, 512, 1024]; grids [128, 64, 32, 16])."""

    H = (128, 64, 32, 16)

    def __init__(self, *, dropout: float = 0.1) -> None:
        super().__init__()
        dim_in = [128, 256, 512, 1024]
        dim_out = [256, 512, 1024, 1024]
        self.dim_in, self.dim_out = dim_in, dim_out
        self.linears = nn.ModuleList([nn.Linear(dim_in[i], dim_out[i]) for i in range(4)])
        self.attentions = nn.ModuleList([
            Transformer(repeat=1, upsample=False, sr_ratio=1, dim=512, nhead=1, dropout=dropout),
            Transformer(repeat=1, upsample=True, sr_ratio=2, dim=512, nhead=8, dropout=dropout),
            Transformer(repeat=2, upsample=True, sr_ratio=2, dim=512, nhead=8, dropout=dropout),
            Transformer(repeat=3, upsample=True, sr_ratio=2, dim=512, nhead=8, dropout=dropout),
        ])
        self.linears2 = nn.ModuleList([nn.Linear(dim_out[i], 512) for i in range(4)])
        assign_rng_names(self)   # attentions.i.trans.r.* instead of four times trans.r.*

    def _params_for_version(self):
        return [m.weight for m in self.linears] + [m.weight for m in self.linears2]

    def _build_shadows(self, device):
        s, e = {}, []
        for grp, mods in (("lin", self.linears), ("lin2", self.linears2)):
            for i, m in enumerate(mods):
                N, K = m.weight.shape
                s[f"{grp}{i}"] = torch.empty(N, K, dtype=torch.bfloat16, device=device)
                s[f"{grp}{i}T"] = torch.empty(K, N, dtype=torch.bfloat16, device=device)
                e.append((m.weight, s[f"{grp}{i}"], s[f"{grp}{i}T"]))
        return s, e

    def _fwd(self, xs, save):
        if len(xs) != 4:
            raise ValueError("lc2is_amd ftn.Decoder: expects the 4 backbone stages")
        require_cuda(xs[0], "x")
        s = self._ensure_ready()
        B = xs[0].shape[0]
        x32, x16 = [], []
        for i, x in enumerate(xs):
            P, C = self.H[i] ** 2, self.dim_in[i]
            if tuple(x.shape) != (B, P, C):
                raise ValueError(f"lc2is_amd ftn.Decoder: stage {i} must be [B, {P}, {C}], got {tuple(x.shape)}")
            t = x.reshape(B * P, C).float().contiguous()
            x32.append(t)
            x16.append(ops.cast_bf16(t))
        # out[i] = linears[i](x[i]) (+ bilinear x2 of the raw next stage for i = 1, 2)   (model/ftn.py:107-123)
        out16 = []
        for i in range(4):
            add = None
            if i in (1, 2):
                add, _ = ops.bilinear_up_fwd(x32[i + 1], B, self.H[i + 1], self.H[i + 1], 2)
            ob, _, _ = ops.gemm_nt(x16[i], s[f"lin{i}"], self.linears[i].bias, resid=add)
            out16.append(ob)
        _, end0, _ = ops.gemm_nt(out16[0], s["lin20"], self.linears2[0].bias, out_bf16=None, out_f32=True)
        ends, sv_att, e16 = [end0], [None], [None]
        for i in range(1, 4):
            eb, ef, _ = ops.gemm_nt(out16[i], s[f"lin2{i}"], self.linears2[i].bias, out_bf16=True, out_f32=True)
            y32, _, sva, cur = self.attentions[i]._fwd(ef, eb, B, self.H[i] ** 2, self.H[i], save)
            if cur != self.H[0] ** 2:
                raise RuntimeError("lc2is_amd ftn.Decoder: stage outputs must all reach 16384 tokens")
            ends.append(y32)
            sv_att.append(sva)
        out, _ = ops.add_n(ends)
        saved = dict(B=B, x16=x16, out16=out16, att=sv_att) if save else None
        return out.view(B, self.H[0] ** 2, 512), saved

    def _bwd(self, gout, saved):
        s = self._sh
        B, x16, out16 = saved["B"], saved["x16"], saved["out16"]
        g32 = gout.reshape(-1, 512).float().contiguous()
        dx = [None] * 4
        up = None   # gradient arriving at x[i] from stage i-1's `add` term
        for i in range(4):
            if i == 0:
                d16 = ops.cast_bf16(g32)
            else:
                d16 = ops.cast_bf16(self.attentions[i]._bwd(g32, saved["att"][i], B))
            linear_bwd_params(d16, out16[i], self.linears2[i].weight, self.linears2[i].bias)
            need_f32 = i in (1, 2)
            dob, dof, _ = ops.gemm_nt(d16, s[f"lin2{i}T"], None, out_bf16=True, out_f32=True if need_f32 else None)
            linear_bwd_params(dob, x16[i], self.linears[i].weight, self.linears[i].bias)
            _, dx[i], _ = ops.gemm_nt(dob, s[f"lin{i}T"], None, resid=up, out_bf16=None, out_f32=True)
            up = None
            if need_f32:   # d(add_i) = d(out_i): back through the bilinear x2 into the raw stage i+1
                up, _ = ops.bilinear_up_bwd(dof, B, self.H[i + 1], self.H[i + 1], 2)
        self._grads_ready()
        return [dx[i].view(B, self.H[i] ** 2, self.dim_in[i]) for i in range(4)]

    def forward(self, x):
        anchor = self.linears2[0].weight
        save = torch.is_grad_enabled() and (anchor.requires_grad or any(t.requires_grad for t in x))
        return _DecoderFn.apply(x[0], x[1], x[2], x[3], anchor, self, save)
