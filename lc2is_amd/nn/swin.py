"""Swin backbone on MI355X — drop-in for ``SwinTransformer`` (reference model/encoder.py:121-131), which wraps
``hf:SwinModel`` (modeling_swin.py) and returns ``hidden_states[:4]``: the patch-embedding output and the outputs of
stages 1-3 AFTER their patch merging — at 512^2 ``[B,16384,96] [B,4096,192] [B,1024,384] [B,256,768]`` (Swin-small),
the four tensors the hierarchical decoders consume.  Stage 4 and the final LayerNorm never reach
those outputs: their parameters exist (checkpoint interchange) but are not evaluated and receive no gradient.

Parameter names are transformers-5.x's (``encoder.embeddings.patch_embeddings.projection.weight``,
``encoder.encoder.layers.{s}.blocks.{b}.attention.{q,k,v,o}_proj.*``, ``...relative_position_bias.
relative_position_bias_table``, ``...layernorm_before/after``, ``...mlp.fc1/fc2``, ``...downsample.{reduction,norm}``);
``load_state_dict`` also accepts the 4.x names (``attention.self.{query,key,value}``, ``attention.output.dense``,
``intermediate.dense``, ``output.dense``) the published checkpoints use.

HIP design: tokens as rows of an fp32 residual stream; per block  LN -> index-map row gather (pad + cyclic shift +
window partition in one pass) -> fused QKV MFMA GEMM -> window attention kernel (relative-position bias + region mask
in-kernel) -> output GEMM -> index-map gather back fused with the residual add -> LN -> fc1+GELU(erf) -> fc2+residual.
Channel counts that are not multiples of 64 (Swin-small/tiny stage 1: 96, 3x96) are zero-padded in the bf16 operand
buffers and weight shadows.  Drop-path (the config default drop_path_rate = 0.1, SwinDropPath on the attention branch) runs in
training mode from per-sample counter-based decisions (no stored mask); hidden / attention dropout are 0 in the config
(hf SwinConfig defaults) and are not modelled.  The whole backbone is ONE autograd node.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import nn

from .. import ops
from .base import DropoutRng, HipModule, WgradBatch, grad_buf, linear_bwd_params, require_cuda, vec_grad


@dataclass(frozen=True)
class SwinArch:
    embed_dim: int = 96
    depths: tuple = (2, 2, 18, 2)
    num_heads: tuple = (3, 6, 12, 24)
    window: int = 7
    patch: int = 4
    eps: float = 1e-5
    mlp_ratio: int = 4
    drop_path_rate: float = 0.1       # hf SwinConfig default


SWIN_T = SwinArch(96, (2, 2, 6, 2), (3, 6, 12, 24))
SWIN_S = SwinArch(96, (2, 2, 18, 2), (3, 6, 12, 24))      # microsoft/swin-small-patch4-window7-224 (model/encoder.py:125)
SWIN_B = SwinArch(128, (2, 2, 18, 2), (4, 8, 16, 32))     # microsoft/swin-base-patch4-window7-224-in22k (model/ftn.py:12)


def _r64(n: int) -> int:
    return (n + 63) // 64 * 64


def relative_position_index(ws: int) -> torch.Tensor:
    """modeling_swin.py:350-365, flattened [ws^2 * ws^2]."""
    coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1).view(-1)


# ---- parameter tree under the reference's / transformers' names ---------------------------------------------------
class _RelBias(nn.Module):
    def __init__(self, nH: int, ws: int):
        super().__init__()
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), nH))


class _Attn(nn.Module):
    def __init__(self, C: int, nH: int, ws: int):
        super().__init__()
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = (nn.Linear(C, C) for _ in range(4))
        self.relative_position_bias = _RelBias(nH, ws)


class _MLP(nn.Module):
    def __init__(self, C: int, ratio: int):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(C, ratio * C), nn.Linear(ratio * C, C)


class _Block(nn.Module):
    def __init__(self, C: int, nH: int, a: SwinArch):
        super().__init__()
        self.attention = _Attn(C, nH, a.window)
        self.layernorm_before = nn.LayerNorm(C, eps=a.eps)
        self.layernorm_after = nn.LayerNorm(C, eps=a.eps)
        self.mlp = _MLP(C, a.mlp_ratio)


class _Merge(nn.Module):
    def __init__(self, C: int):
        super().__init__()
        self.reduction = nn.Linear(4 * C, 2 * C, bias=False)
        self.norm = nn.LayerNorm(4 * C)


class _Stage(nn.Module):
    def __init__(self, C: int, depth: int, nH: int, a: SwinArch, merge: bool):
        super().__init__()
        self.blocks = nn.ModuleList([_Block(C, nH, a) for _ in range(depth)])
        self.downsample = _Merge(C) if merge else None


class _Encoder(nn.Module):
    def __init__(self, a: SwinArch):
        super().__init__()
        n = len(a.depths)
        self.layers = nn.ModuleList([_Stage(a.embed_dim << i, a.depths[i], a.num_heads[i], a, i < n - 1) for i in range(n)])


class _PatchEmb(nn.Module):
    def __init__(self, a: SwinArch):
        super().__init__()
        self.projection = nn.Conv2d(3, a.embed_dim, kernel_size=a.patch, stride=a.patch)


class _Embeddings(nn.Module):
    def __init__(self, a: SwinArch):
        super().__init__()
        self.patch_embeddings = _PatchEmb(a)
        self.norm = nn.LayerNorm(a.embed_dim)


class _SwinModel(nn.Module):
    def __init__(self, a: SwinArch):
        super().__init__()
        self.embeddings = _Embeddings(a)
        self.encoder = _Encoder(a)
        self.layernorm = nn.LayerNorm(a.embed_dim << (len(a.depths) - 1), eps=a.eps)
        for m in self.modules():      # SwinPreTrainedModel._init_weights: normal(0, 0.02), LN 1/0, bias tables 0
            if isinstance(m, (nn.Linear, nn.Conv2d)):
                nn.init.normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)


_LEGACY = ((".attention.self.query.", ".attention.q_proj."), (".attention.self.key.", ".attention.k_proj."),
           (".attention.self.value.", ".attention.v_proj."), (".attention.output.dense.", ".attention.o_proj."),
           (".attention.self.relative_position_bias_table", ".attention.relative_position_bias.relative_position_bias_table"),
           (".intermediate.dense.", ".mlp.fc1."), (".output.dense.", ".mlp.fc2."))


class _SwinFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pixel_values, anchor, mod, save):
        outs, saved = mod._fwd(pixel_values, save)
        ctx.mod, ctx.saved = mod, saved
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        ctx.mod._bwd(gouts, ctx.saved)
        ctx.saved = None
        return None, None, None, None


class SwinTransformer(HipModule):
    """Drop-in for model/encoder.py:121-131.  ``forward(pixel_values [B,3,H,W]) -> (h0, h1, h2, h3)``."""

    N_OUT = 4

    def __init__(self, arch: SwinArch | None = None, *, drop_path_rate: float | None = None) -> None:
        super().__init__()
        a = arch or SWIN_S
        if drop_path_rate is not None:
            a = SwinArch(a.embed_dim, a.depths, a.num_heads, a.window, a.patch, a.eps, a.mlp_ratio, drop_path_rate)
        if len(a.depths) != 4 or a.window * a.window > 64 or any((a.embed_dim << i) != 32 * a.num_heads[i] for i in range(4)):
            raise NotImplementedError("lc2is_amd SwinTransformer: 4 stages, head_dim 32, window <= 8 (every published Swin)")
        if a.embed_dim % 32:
            raise NotImplementedError("lc2is_amd SwinTransformer: embed_dim must be a multiple of 32")
        self.arch = a
        self.encoder = _SwinModel(a)
        self._maps = {}
        self._onehot = None
        self._register_load_state_dict_pre_hook(self._remap_legacy_keys)

    @staticmethod
    def _remap_legacy_keys(state_dict, prefix, *args):
        for k in list(state_dict):
            if not k.startswith(prefix):
                continue
            if k.endswith("relative_position_index"):
                state_dict.pop(k)
                continue
            nk = k
            for old, new in _LEGACY:      # ordered: ".attention.output.dense." is rewritten before the MLP's ".output.dense."
                nk = nk.replace(old, new)
            if nk != k:
                state_dict[nk] = state_dict.pop(k)

    # ---- shadows ----------------------------------------------------------------------------------------------------
    def _stages(self):
        return list(self.encoder.encoder.layers)[: self.N_OUT - 1]

    def _build_shadows(self, device):
        a = self.arch
        bf = dict(dtype=torch.bfloat16, device=device)
        entries, stages = [], []
        k = 3 * a.patch ** 2
        kpad = _r64(k)
        wp = torch.zeros(a.embed_dim, kpad, **bf)
        entries.append((self.encoder.embeddings.patch_embeddings.projection.weight.view(a.embed_dim, k), wp[:, :k], None))
        for si, st in enumerate(self._stages()):
            C = a.embed_dim << si
            Cp, Q3p, F = _r64(C), _r64(3 * C), a.mlp_ratio * C
            blocks = []
            for b in st.blocks:
                at = b.attention
                s = dict(wqkv=torch.zeros(Q3p, Cp, **bf), wqkvT=torch.zeros(Cp, Q3p, **bf),
                         bqkv=torch.zeros(Q3p, dtype=torch.float32, device=device),
                         wo=torch.zeros(C, Cp, **bf), woT=torch.zeros(C, Cp, **bf),
                         w1=torch.zeros(F, Cp, **bf), w1T=torch.zeros(C, F, **bf),
                         w2=torch.zeros(C, F, **bf), w2T=torch.zeros(F, Cp, **bf), bias=None)
                for j, lin in enumerate((at.q_proj, at.k_proj, at.v_proj)):
                    entries.append((lin.weight, s["wqkv"][j * C:(j + 1) * C, :C], s["wqkvT"][:C, j * C:(j + 1) * C]))
                    entries.append((lin.bias, s["bqkv"][j * C:(j + 1) * C], None))
                entries.append((at.o_proj.weight, s["wo"][:, :C], s["woT"][:, :C]))
                entries.append((b.mlp.fc1.weight, s["w1"][:, :C], s["w1T"]))
                entries.append((b.mlp.fc2.weight, s["w2"], s["w2T"][:, :C]))
                blocks.append(s)
            red = st.downsample.reduction.weight
            wr, wrT = torch.empty(2 * C, 4 * C, **bf), torch.empty(4 * C, 2 * C, **bf)
            entries.append((red, wr, wrT))
            stages.append(dict(blocks=blocks, wred=wr, wredT=wrT))
        return dict(wp=wp, kpad=kpad, stages=stages), entries

    def _post_refresh(self):
        """Expand every block's relative-position table to the [nH, S, S] bias the attention kernel reads."""
        ws = self.arch.window
        S = ws * ws
        dev = self._sh["wp"].device
        idx = getattr(self, "_rel_idx", None)
        if idx is None or idx.device != dev:      # built once: no host-to-device copy per refresh (hipGraph-capturable)
            idx = self._rel_idx = relative_position_index(ws).to(dev)
        if self._onehot is None or self._onehot[0].device != dev:
            # inverse of the index (init-time host code): for every table row t, the (query, key) pairs that read it
            T = (2 * ws - 1) ** 2
            host = relative_position_index(ws)
            order = torch.sort(host, stable=True).indices                   # pairs grouped by t, ascending pair id inside
            offs = torch.zeros(T + 1, dtype=torch.int64)
            offs[1:] = torch.cumsum(torch.bincount(host, minlength=T), 0)
            self._onehot = (offs.to(torch.int32).to(dev), order.to(torch.int32).to(dev))
        for st, ss in zip(self._stages(), self._sh["stages"]):
            for b, s in zip(st.blocks, ss["blocks"]):
                t = b.attention.relative_position_bias.relative_position_bias_table.detach()
                s["bias"] = t[idx].view(S, S, -1).permute(2, 0, 1).contiguous()

    # ---- index maps ---------------------------------------------------------------------------------------------------
    def _window_maps(self, B, H, W, shift, dev):
        ws = self.arch.window
        key = ("win", B, H, W, shift, str(dev))
        m = self._maps.get(key)
        if m is None:
            Hp, Wp = (H + ws - 1) // ws * ws, (W + ws - 1) // ws * ws
            py = torch.arange(Hp, device=dev).view(Hp, 1).expand(Hp, Wp)
            px = torch.arange(Wp, device=dev).view(1, Wp).expand(Hp, Wp)
            sy, sx = (py + shift) % Hp, (px + shift) % Wp              # torch.roll(-shift): shifted[p] = padded[p + shift]
            valid = (sy < H) & (sx < W)
            tok = torch.where(valid, sy * W + sx, torch.full_like(sy, -1))
            win = tok.view(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1)     # window-major order
            off = (torch.arange(B, device=dev) * (H * W)).view(B, 1)
            fwd = torch.where(win.view(1, -1) >= 0, win.view(1, -1) + off, torch.full_like(win.view(1, -1).expand(B, -1), -1))
            fwd = fwd.reshape(-1).to(torch.int32).contiguous()
            inv = torch.full((B * H * W,), -1, dtype=torch.int32, device=dev)
            rows = torch.arange(fwd.numel(), device=dev, dtype=torch.int32)
            ok = fwd >= 0
            inv[fwd[ok].long()] = rows[ok]
            m = dict(fwd=fwd, inv=inv, Hp=Hp, Wp=Wp, nwx=Wp // ws, per_img=(Hp // ws) * (Wp // ws), nwin=B * (Hp // ws) * (Wp // ws))
            self._maps[key] = m
        return m

    def _merge_maps(self, B, H, W, dev):
        key = ("merge", B, H, W, str(dev))
        m = self._maps.get(key)
        if m is None:
            H2, W2 = (H + 1) // 2, (W + 1) // 2
            oy = torch.arange(H2, device=dev).view(H2, 1, 1)
            ox = torch.arange(W2, device=dev).view(1, W2, 1)
            part = torch.arange(4, device=dev).view(1, 1, 4)          # concat order: (row0,col0), (row1,col0), (row0,col1), (row1,col1)
            sy, sx = 2 * oy + part % 2, 2 * ox + part // 2
            valid = (sy < H) & (sx < W)
            tok = torch.where(valid, sy * W + sx, torch.full_like(sy + sx, -1)).reshape(1, -1)
            off = (torch.arange(B, device=dev) * (H * W)).view(B, 1)
            fwd = torch.where(tok >= 0, tok + off, torch.full_like(tok.expand(B, -1), -1)).reshape(-1).to(torch.int32).contiguous()
            inv = torch.full((B * H * W,), -1, dtype=torch.int32, device=dev)
            rows = torch.arange(fwd.numel(), device=dev, dtype=torch.int32)
            ok = fwd >= 0
            inv[fwd[ok].long()] = rows[ok]
            m = dict(fwd=fwd, inv=inv, H2=H2, W2=W2)
            self._maps[key] = m
        return m

    # ---- helpers --------------------------------------------------------------------------------------------------------
    @staticmethod
    def _padded(M, C, Cp, dev):
        if Cp == C:
            return torch.empty(M, C, dtype=torch.bfloat16, device=dev)
        return torch.zeros(M, Cp, dtype=torch.bfloat16, device=dev)

    def _cast_padded(self, x32, Cp):
        M, C = x32.shape
        buf = self._padded(M, C, Cp, x32.device)
        ops.cast_bf16(x32, buf[:, :C])
        return buf

    @staticmethod
    def _wgrad_padded(dy16, x16p, weight, bias, C_in):
        """Linear(C_in -> N) parameter grads when the activation operand is zero-padded to a multiple of 64 columns."""
        if x16p.shape[1] == C_in:
            linear_bwd_params(dy16, x16p, weight, bias)
            return
        gb, accb = vec_grad(bias)
        if weight.requires_grad:
            gw, acc = grad_buf(weight)
            tmp_b = torch.empty_like(gb) if gb is not None else None
            tmp = ops.gemm_tn(dy16, x16p, db=tmp_b)                     # bias gradient rides on the same launch
            gw.add_(tmp[:, :C_in]) if acc else gw.copy_(tmp[:, :C_in])
            if gb is not None:
                gb.add_(tmp_b) if accb else gb.copy_(tmp_b)
        elif gb is not None:
            ops.colsum(dy16, gb, accumulate=accb)

    @staticmethod
    def _qkv_wgrad(dqkv16, win16p, at: _Attn, C):
        """q, k and v share their input: one [3C, Cp] weight-gradient GEMM (+ fused bias gradient), then three slices."""
        ps = (at.q_proj, at.k_proj, at.v_proj)
        if win16p.shape[1] == C or not all(l.weight.requires_grad and l.bias.requires_grad for l in ps):
            # unpadded channels (stages 2-4): three strided problems of the block's grouped launch, written in place
            for j, lin in enumerate(ps):
                SwinTransformer._wgrad_padded(dqkv16[:, j * C:(j + 1) * C], win16p, lin.weight, lin.bias, C)
            return
        tmp_b = torch.empty(3 * C, dtype=torch.float32, device=dqkv16.device)
        tmp = ops.gemm_tn(dqkv16, win16p, db=tmp_b)                       # [3C, Cp]
        for j, lin in enumerate(ps):
            gw, acc = grad_buf(lin.weight)
            gb, accb = grad_buf(lin.bias)
            w, b = tmp[j * C:(j + 1) * C, :C], tmp_b[j * C:(j + 1) * C]
            gw.add_(w) if acc else gw.copy_(w)
            gb.add_(b) if accb else gb.copy_(b)

    # ---- one block ------------------------------------------------------------------------------------------------------
    def _block_fwd(self, x32, blk: _Block, s, ctx, shift, save, dp=None):
        a = self.arch
        B, H, W, C, nH = ctx["B"], ctx["H"], ctx["W"], ctx["C"], ctx["nH"]
        Cp, ws = _r64(C), a.window
        M = B * H * W
        mp = self._window_maps(B, H, W, shift, x32.device)
        h16 = self._padded(M, C, Cp, x32.device)
        _, _, m1, r1 = ops.layernorm_fwd(x32, blk.layernorm_before.weight, blk.layernorm_before.bias, a.eps, save_stats=save,
                                         out_bf16=h16[:, :C])
        win16 = ops.rows_gather(h16, mp["fwd"])                                                  # [Mw, Cp]
        qkv, _, _ = ops.gemm_nt(win16, s["wqkv"], s["bqkv"])                                      # [Mw, Q3p]
        o16p = self._padded(qkv.shape[0], C, Cp, x32.device)
        _, lse = ops.swin_attn_fwd(qkv[:, :3 * C], s["bias"], mp["nwin"], mp["per_img"], mp["nwx"], mp["Hp"], mp["Wp"], ws,
                                   shift, nH, 32 ** -0.5, save_lse=save, out=o16p[:, :C])
        _, y32, _ = ops.gemm_nt(o16p, s["wo"], blk.attention.o_proj.bias, out_bf16=None, out_f32=True)   # [Mw, C]
        if dp is not None:   # SwinDropPath on the attention branch (modeling_swin.py:567): one decision per SAMPLE; in window
            ops.dropout_rows_f32(y32, dp[0], dp[1], out_f32=y32, rows_per_sample=mp["per_img"] * ws * ws)   # order a sample
        x_mid = ops.rows_gather(y32, mp["inv"], add=x32)          # owns per_img windows.  un-window + residual
        h2 = self._padded(M, C, Cp, x32.device)
        _, _, m2, r2 = ops.layernorm_fwd(x_mid, blk.layernorm_after.weight, blk.layernorm_after.bias, a.eps, save_stats=save,
                                         out_bf16=h2[:, :C])
        act, _, z = ops.gemm_nt(h2, s["w1"], blk.mlp.fc1.bias, act=ops.ACT_GELU_ERF, aux_out=True if save else None)
        _, x_out, _ = ops.gemm_nt(act, s["w2"], blk.mlp.fc2.bias, resid=x_mid, out_bf16=None, out_f32=True)
        sv = dict(x=x32, m1=m1, r1=r1, win16=win16, qkv=qkv, o16p=o16p, lse=lse, x_mid=x_mid, m2=m2, r2=r2, h2=h2, z=z,
                  act=act, shift=shift, dp=dp) if save else None
        return x_out, sv

    def _block_bwd(self, g32, blk: _Block, s, ctx, sv):
        with WgradBatch():   # the block's weight gradients (q, k, v, o, fc1, fc2) leave as one grouped grid + one ordered reduce
            return self._block_bwd_inner(g32, blk, s, ctx, sv)

    def _block_bwd_inner(self, g32, blk: _Block, s, ctx, sv):
        a = self.arch
        B, H, W, C, nH = ctx["B"], ctx["H"], ctx["W"], ctx["C"], ctx["nH"]
        Cp, ws, shift = _r64(C), a.window, sv["shift"]
        mp = self._window_maps(B, H, W, shift, g32.device)
        at, mlp = blk.attention, blk.mlp
        # x_out = x_mid + fc2(gelu(fc1(LN_after(x_mid))))
        g16 = self._cast_padded(g32, Cp)
        linear_bwd_params(g16[:, :C], sv["act"], mlp.fc2.weight, mlp.fc2.bias)
        dz, _, _ = ops.gemm_nt(g16, s["w2T"], None, act=ops.ACT_DGELU_ERF, aux_in=sv["z"])         # [M, F]
        self._wgrad_padded(dz, sv["h2"], mlp.fc1.weight, mlp.fc1.bias, C)
        dh2, _, _ = ops.gemm_nt(dz, s["w1T"], None)                                               # [M, C]
        dg, accg = vec_grad(blk.layernorm_after.weight)
        db, _ = vec_grad(blk.layernorm_after.bias)
        gm32, _, _, _ = ops.layernorm_bwd(dh2, sv["x_mid"], blk.layernorm_after.weight, sv["m2"], sv["r2"], dres=g32, dgamma=dg,
                                          dbeta=db, accumulate=accg, want_bf16=False, need_param_grads=dg is not None)
        # x_mid = x + unwindow(o_proj(attn(qkv(window(LN_before(x))))))
        Mw = mp["fwd"].numel()
        dy16 = self._padded(Mw, C, Cp, g32.device)
        gbr32 = gm32
        if sv["dp"] is not None:   # gradient wrt the drop-path'ed attention branch: the forward's per-sample decisions again
            gbr32, _ = ops.dropout_rows_f32(gm32, sv["dp"][0], sv["dp"][1], rows_per_sample=H * W)
        ops.rows_gather(gbr32, mp["fwd"], out=dy16, cols=C)                                       # pad rows -> 0
        self._wgrad_padded(dy16[:, :C], sv["o16p"], at.o_proj.weight, at.o_proj.bias, C)
        do16, _, _ = ops.gemm_nt(dy16, s["woT"], None)                                            # [Mw, C]
        S = ws * ws
        table = at.relative_position_bias.relative_position_bias_table
        dbias = torch.empty(nH, S, S, dtype=torch.float32, device=g32.device) if table.requires_grad else None
        qkv = sv["qkv"]
        dqkv = self._padded(qkv.shape[0], 3 * C, qkv.shape[1], g32.device)
        ops.swin_attn_bwd(qkv[:, :3 * C], sv["o16p"][:, :C], do16, sv["lse"], s["bias"], mp["nwin"], mp["per_img"], mp["nwx"],
                          mp["Hp"], mp["Wp"], ws, shift, nH, 32 ** -0.5, dbias=dbias, dqkv=dqkv[:, :3 * C])
        if dbias is not None:
            gt, acc = grad_buf(table)
            ops.swin_bias_table_grad(dbias, self._onehot[0], self._onehot[1], gt, accumulate=acc)   # [T, nH], fixed order
        self._qkv_wgrad(dqkv[:, :3 * C], sv["win16"], at, C)
        dwin16, _, _ = ops.gemm_nt(dqkv, s["wqkvT"][:C], None)                                     # [Mw, C]
        dh16 = ops.rows_gather(dwin16, mp["inv"])                                                 # [M, C] bf16
        dg, accg = vec_grad(blk.layernorm_before.weight)
        db, _ = vec_grad(blk.layernorm_before.bias)
        gx32, _, _, _ = ops.layernorm_bwd(dh16, sv["x"], blk.layernorm_before.weight, sv["m1"], sv["r1"], dres=gm32, dgamma=dg,
                                          dbeta=db, accumulate=accg, want_bf16=False, need_param_grads=dg is not None)
        return gx32

    # ---- patch merging ----------------------------------------------------------------------------------------------------
    def _merge_fwd(self, x32, mg: _Merge, ss, ctx, save):
        B, H, W, C = ctx["B"], ctx["H"], ctx["W"], ctx["C"]
        mp = self._merge_maps(B, H, W, x32.device)
        M2 = B * mp["H2"] * mp["W2"]
        m32 = ops.rows_gather(x32, mp["fwd"]).view(M2, 4 * C)
        mn16, _, mm, rm = ops.layernorm_fwd(m32, mg.norm.weight, mg.norm.bias, mg.norm.eps, save_stats=save)
        _, y32, _ = ops.gemm_nt(mn16, ss["wred"], None, out_bf16=None, out_f32=True)              # [M2, 2C]
        return y32, (dict(m32=m32, mm=mm, rm=rm, mn16=mn16) if save else None)

    def _merge_bwd(self, g32, gskip, mg: _Merge, ss, ctx, sv):
        B, H, W, C = ctx["B"], ctx["H"], ctx["W"], ctx["C"]
        mp = self._merge_maps(B, H, W, g32.device)
        g16 = ops.cast_bf16(g32)
        linear_bwd_params(g16, sv["mn16"], mg.reduction.weight, None)
        dmn16, _, _ = ops.gemm_nt(g16, ss["wredT"], None)                                         # [M2, 4C]
        dg, accg = vec_grad(mg.norm.weight)
        db, _ = vec_grad(mg.norm.bias)
        dm32, _, _, _ = ops.layernorm_bwd(dmn16, sv["m32"], mg.norm.weight, sv["mm"], sv["rm"], dgamma=dg, dbeta=db,
                                          accumulate=accg, want_bf16=False, need_param_grads=dg is not None)
        return ops.rows_gather(dm32.view(-1, C), mp["inv"], add=gskip)                            # [M, C] (+ this level's own grad)

    # ---- whole backbone ----------------------------------------------------------------------------------------------------
    def _fwd(self, pixel_values, save):
        require_cuda(pixel_values, "pixel_values")
        a = self.arch
        B, ch, Hi, Wi = pixel_values.shape
        if ch != 3 or Hi != Wi or Hi % a.patch:
            raise ValueError("lc2is_amd SwinTransformer: expects square [B,3,H,W] images with H divisible by the patch size")
        G = Hi // a.patch
        if G // 4 <= a.window:
            raise NotImplementedError("lc2is_amd SwinTransformer: stage-3 grids no larger than the window are outside the path")
        sh = self._ensure_ready()
        emb = self.encoder.embeddings
        cols = ops.patchify(pixel_values.float().contiguous(), a.patch, sh["kpad"])
        _, pe, _ = ops.gemm_nt(cols, sh["wp"], emb.patch_embeddings.projection.bias, out_bf16=None, out_f32=True)
        _, x, m0, r0 = ops.layernorm_fwd(pe, emb.norm.weight, emb.norm.bias, emb.norm.eps, save_stats=save, out_bf16=None,
                                         out_f32=True)
        outs = [x.view(B, G * G, a.embed_dim)]
        saved = dict(cols=cols, pe=pe, m0=m0, r0=r0, B=B, G=G, stages=[]) if save else None
        H = W = G
        for si, (st, ss) in enumerate(zip(self._stages(), sh["stages"])):
            C = a.embed_dim << si
            ctx = dict(B=B, H=H, W=W, C=C, nH=a.num_heads[si])
            svb = []
            for bi, (blk, s) in enumerate(zip(st.blocks, ss["blocks"])):
                # stochastic depth: rate rises linearly over ALL blocks of the four stages (modeling_swin.py:758)
                rate = a.drop_path_rate * (sum(a.depths[:si]) + bi) / max(sum(a.depths) - 1, 1)
                dp = (rate, DropoutRng.next_seed(f"swin.{si}.{bi}.drop_path", rate)) if (self.training and rate > 0) else None
                x, sv = self._block_fwd(x, blk, s, ctx, 0 if bi % 2 == 0 else a.window // 2, save, dp)
                svb.append(sv)
            x, svm = self._merge_fwd(x, st.downsample, ss, ctx, save)
            H, W = (H + 1) // 2, (W + 1) // 2
            outs.append(x.view(B, H * W, 2 * C))
            if save:
                saved["stages"].append(dict(ctx=ctx, blocks=svb, merge=svm))
        return outs, saved

    def _bwd(self, gouts, saved):
        a, sh = self.arch, self._sh
        B, G = saved["B"], saved["G"]
        dev = saved["pe"].device

        def grad_of(i, rows, C):
            g = gouts[i]
            return None if g is None else g.reshape(rows, C).float().contiguous()

        n = self.N_OUT - 1
        last = saved["stages"][-1]["ctx"]
        H2, W2 = (last["H"] + 1) // 2, (last["W"] + 1) // 2
        g = grad_of(n, B * H2 * W2, 2 * last["C"])
        if g is None:
            g = torch.zeros(B * H2 * W2, 2 * last["C"], dtype=torch.float32, device=dev)
        for si in reversed(range(n)):
            st, ss, sv = self._stages()[si], sh["stages"][si], saved["stages"][si]
            ctx = sv["ctx"]
            gskip = grad_of(si, B * ctx["H"] * ctx["W"], ctx["C"])       # gradient of hidden_states[si] itself
            # hidden_states[si] is the INPUT of stage si: its own gradient joins after the stage's blocks
            g = self._merge_bwd(g, None, st.downsample, ss, ctx, sv["merge"])
            for blk, s, svb in zip(reversed(list(st.blocks)), reversed(ss["blocks"]), reversed(sv["blocks"])):
                g = self._block_bwd(g, blk, s, ctx, svb)
            if gskip is not None:
                g = g + gskip
        emb = self.encoder.embeddings
        dg, accg = vec_grad(emb.norm.weight)
        db, _ = vec_grad(emb.norm.bias)
        _, dpe16, _, _ = ops.layernorm_bwd(g, saved["pe"], emb.norm.weight, saved["m0"], saved["r0"], dgamma=dg, dbeta=db,
                                           accumulate=accg, want_f32=False, need_param_grads=dg is not None)
        proj = emb.patch_embeddings.projection
        k = 3 * a.patch ** 2
        if proj.weight.requires_grad:
            gw, acc = grad_buf(proj.weight)
            tmp = ops.gemm_tn(dpe16, saved["cols"])
            gw.view(a.embed_dim, k).add_(tmp[:, :k]) if acc else gw.view(a.embed_dim, k).copy_(tmp[:, :k])
        if proj.bias.requires_grad:
            gb, accb = grad_buf(proj.bias)
            ops.colsum(dpe16, gb, accumulate=accb)
        self._grads_ready()

    def forward(self, pixel_values: torch.Tensor):
        anchor = self.encoder.embeddings.norm.weight
        save = torch.is_grad_enabled() and anchor.requires_grad
        return _SwinFn.apply(pixel_values, anchor, self, save)
