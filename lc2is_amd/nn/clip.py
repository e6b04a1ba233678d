"""CLIP image / text encoders on MI355X — drop-ins for the reference's ``model/encoder.py`` wrappers.

Reference interface mirrored here:
  ImageEncoderCLIP(in_size, patch_size=16).forward(pixel_values) -> [B,P,C]          model/encoder.py:11-47
  ImageEncoderCLIPFull(...).forward(pixel_values) -> [B,P+1,C]                        model/encoder.py:49-85
  TextEncoderCLIP(patch_size=16).forward(input_ids, attention_mask) -> [B,L,C]        model/encoder.py:87-102
  TextEncoderCLIPPooler(...).forward(input_ids, attention_mask=None) -> [B,C]         model/encoder.py:104-119
  .hidden_size() -> int
Parameters live under ``enc.*`` with transformers-5.15 names (``embeddings.class_embedding``,
``pre_layrnorm`` [sic], ``encoder.layers.{i}.self_attn.{q,k,v,out}_proj`` ...); the legacy
``enc.vision_model.`` / ``enc.text_model.`` prefixes are accepted on load.

The reference constructs these through ``from_pretrained(<hub name>)``; there is no network here, so the
constructors build the same architecture with random init and weights arrive via ``load_state_dict``.

Compute: the whole layer stack runs on the HIP kernels of ``liblc2is_hip.so`` (bf16 MFMA GEMMs with fused
bias / quick_gelu / residual epilogues, fused attention, LayerNorm over an fp32 residual stream).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
from torch import nn

from .. import ops
from .base import HipModule, WgradBatch, grad_buf, linear_bwd_params, require_cuda, vec_grad


@dataclass
class ClipArch:
    hidden: int = 768
    heads: int = 12
    layers: int = 12
    intermediate: int = 3072
    eps: float = 1e-5
    # text only
    vocab: int = 49408
    max_pos: int = 77
    eos_token_id: int = 49407


VIT_B16 = ClipArch(768, 12, 12, 3072)
VIT_L14 = ClipArch(1024, 16, 24, 4096)
TEXT_B = ClipArch(512, 8, 12, 2048)
TEXT_L = ClipArch(768, 12, 12, 3072)


# ---- parameter containers (names == transformers 5.15 CLIP) -------------------------------------------
class _Attn(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = (nn.Linear(c, c) for _ in range(4))


class _MLP(nn.Module):
    def __init__(self, c, i):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(c, i), nn.Linear(i, c)


class _Layer(nn.Module):
    def __init__(self, a: ClipArch):
        super().__init__()
        self.self_attn = _Attn(a.hidden)
        self.layer_norm1 = nn.LayerNorm(a.hidden, eps=a.eps)
        self.mlp = _MLP(a.hidden, a.intermediate)
        self.layer_norm2 = nn.LayerNorm(a.hidden, eps=a.eps)


class _Stack(nn.Module):
    def __init__(self, a: ClipArch):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(a) for _ in range(a.layers)])


class _VisionEmbeddings(nn.Module):
    def __init__(self, a: ClipArch, image_size: int, patch: int):
        super().__init__()
        self.class_embedding = nn.Parameter(torch.randn(a.hidden))
        self.patch_embedding = nn.Conv2d(3, a.hidden, patch, stride=patch, bias=False)
        self.position_embedding = nn.Embedding((image_size // patch) ** 2 + 1, a.hidden)


class _TextEmbeddings(nn.Module):
    def __init__(self, a: ClipArch):
        super().__init__()
        self.token_embedding = nn.Embedding(a.vocab, a.hidden)
        self.position_embedding = nn.Embedding(a.max_pos, a.hidden)


class _VisionParams(nn.Module):
    def __init__(self, a: ClipArch, image_size: int, patch: int):
        super().__init__()
        self.embeddings = _VisionEmbeddings(a, image_size, patch)
        self.pre_layrnorm = nn.LayerNorm(a.hidden, eps=a.eps)
        self.encoder = _Stack(a)
        self.post_layernorm = nn.LayerNorm(a.hidden, eps=a.eps)  # part of the checkpoint; unused for tokens


class _TextParams(nn.Module):
    def __init__(self, a: ClipArch):
        super().__init__()
        self.embeddings = _TextEmbeddings(a)
        self.encoder = _Stack(a)
        self.final_layer_norm = nn.LayerNorm(a.hidden, eps=a.eps)


def _clip_init(enc: nn.Module, a: ClipArch) -> None:
    """transformers CLIPPreTrainedModel._init_weights, restated (initializer_factor = 1)."""
    in_std = a.hidden ** -0.5 * (2 * a.layers) ** -0.5
    out_std = a.hidden ** -0.5
    fc_std = (2 * a.hidden) ** -0.5
    with torch.no_grad():
        for layer in enc.encoder.layers:
            for n in ("q_proj", "k_proj", "v_proj"):
                getattr(layer.self_attn, n).weight.normal_(0, in_std)
                getattr(layer.self_attn, n).bias.zero_()
            layer.self_attn.out_proj.weight.normal_(0, out_std)
            layer.self_attn.out_proj.bias.zero_()
            layer.mlp.fc1.weight.normal_(0, fc_std)
            layer.mlp.fc2.weight.normal_(0, in_std)
            layer.mlp.fc1.bias.zero_()
            layer.mlp.fc2.bias.zero_()
        emb = enc.embeddings
        if hasattr(emb, "class_embedding"):
            emb.class_embedding.normal_(0, a.hidden ** -0.5)
            emb.patch_embedding.weight.normal_(0, 0.02)
            emb.position_embedding.weight.normal_(0, 0.02)
        else:
            emb.token_embedding.weight.normal_(0, 0.02)
            emb.position_embedding.weight.normal_(0, 0.02)


# ---- the shared pre-LN layer stack -------------------------------------------------------------------
def _stack_shadows(stack: _Stack, a: ClipArch, device):
    """bf16 shadows for one encoder stack: fused QKV [3C,C] (+T), out/fc1/fc2 (+T), fused fp32 QKV bias."""
    C, I = a.hidden, a.intermediate
    bf = dict(dtype=torch.bfloat16, device=device)
    per, entries = [], []
    for layer in stack.layers:
        at, mlp = layer.self_attn, layer.mlp
        s = dict(wqkv=torch.empty(3 * C, C, **bf), wqkvT=torch.empty(C, 3 * C, **bf),
                 bqkv=torch.empty(3 * C, dtype=torch.float32, device=device),
                 wo=torch.empty(C, C, **bf), woT=torch.empty(C, C, **bf),
                 w1=torch.empty(I, C, **bf), w1T=torch.empty(C, I, **bf),
                 w2=torch.empty(C, I, **bf), w2T=torch.empty(I, C, **bf))
        for j, proj in enumerate((at.q_proj, at.k_proj, at.v_proj)):
            entries.append((proj.weight, s["wqkv"][j * C:(j + 1) * C], s["wqkvT"][:, j * C:(j + 1) * C]))
            entries.append((proj.bias, s["bqkv"][j * C:(j + 1) * C], None))
        entries.append((at.out_proj.weight, s["wo"], s["woT"]))
        entries.append((mlp.fc1.weight, s["w1"], s["w1T"]))
        entries.append((mlp.fc2.weight, s["w2"], s["w2T"]))
        per.append(s)
    return per, entries


# The residual stream of the towers: fp32 (default) or bf16 (LC2IS_RESID_STREAM=bf16, round 5).  In bf16 the two residual joins of
# a layer are `bf16(acc + bias + x)` epilogues (fp32 add, ONE rounding; LC2IS_ACT_ADD_AUX) that read 2 and write 2 bytes per element
# instead of 4 + 4, LayerNorm reads 2 instead of 4 (forward and backward), and the saved x / x_mid halve: +1.3 % images/s on the
# headline step (profiles/r05_bench_ab_resid_stream.txt).  It stays OPT-IN because the stream is then rounded to 8 significant bits at
# each of the 2 x layers joins, and every parity margin pays for it (profiles/r05_parity_resid_stream.txt): config-2 logits rel-L2
# 6.4e-3 -> 9.8e-3, the ViT tower's error at depth 12 3.1e-3 -> 8.9e-3, config 4 at its 24 layers 6.6e-3 -> 1.26e-2 (bound 1.3e-2), and
# the decoder's `linear1` gradient (relu masks that flip on noisier inputs) 3.9e-2 -> 9.2e-2, past the stated 8e-2 — parity comes first.
_RESID_BF16 = __import__('os').environ.get("LC2IS_RESID_STREAM", "f32") == "bf16"


def _stack_fwd(x, stack: _Stack, sh, a: ClipArch, B: int, S: int, kbias, causal: bool, save: bool):
    """x [B*S, C] residual stream (bf16 when _RESID_BF16, else fp32) -> (x_out in the same type, saved list).
    hf:modeling_clip.py:362-383 per layer."""
    C, H = a.hidden, a.heads
    D = C // H
    scale = D ** -0.5
    saved = []
    lean = x.dtype == torch.bfloat16
    # round 5: on an fp32 stream the two residual-join GEMMs of a layer also write the LayerNorm that follows them (layer_norm2
    # behind out-proj, the NEXT layer's layer_norm1 behind fc2) — ops.gemm_nt_ln, one launch instead of two (width 768 / 384 at
    # enough rows for the 256x384 tiles to fill the chip: the ViT-B vision tower at B >= 28; every other stack keeps the LayerNorm kernel)
    fuse = (not lean) and ops.gemm_nt_ln_ok(x.shape[0], C, C) and ops.gemm_nt_ln_ok(x.shape[0], C, a.intermediate)
    nxt_ln = None   # (h, mean, rstd) of this layer's layer_norm1 when the previous layer's fc2 launch has produced it
    layers = list(stack.layers)
    for li, (layer, s) in enumerate(zip(layers, sh)):
        if nxt_ln is not None:
            h, m1, r1 = nxt_ln
        else:
            h, _, m1, r1 = ops.layernorm_fwd(x, layer.layer_norm1.weight, layer.layer_norm1.bias, a.eps, save_stats=save)
        qkv, _, _ = ops.gemm_nt(h, s["wqkv"], s["bqkv"])
        o, lse = ops.attention_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, H, S, S, D, scale, causal=causal,
                                   kbias=kbias, save_lse=save)
        if fuse:
            x_mid, h2, m2, r2 = ops.gemm_nt_ln(o, s["wo"], layer.self_attn.out_proj.bias, x, layer.layer_norm2.weight,
                                               layer.layer_norm2.bias, a.eps, save_stats=save)
        else:
            if lean:
                x_mid, _, _ = ops.gemm_nt(o, s["wo"], layer.self_attn.out_proj.bias, resid=x)
            else:
                _, x_mid, _ = ops.gemm_nt(o, s["wo"], layer.self_attn.out_proj.bias, resid=x, out_bf16=None, out_f32=True)
            h2, _, m2, r2 = ops.layernorm_fwd(x_mid, layer.layer_norm2.weight, layer.layer_norm2.bias, a.eps,
                                              save_stats=save)
        # (codes 5/6 — save quick_gelu'(z), multiply in the backward — exist in the C ABI; the pair measured 4 % slower
        #  end to end than saving z: the forward variant's sigmoid temporaries spill beside the 128 accumulators)
        act, _, z = ops.gemm_nt(h2, s["w1"], layer.mlp.fc1.bias, act=ops.ACT_QUICK_GELU, aux_out=True if save else None)
        nxt_ln = None
        if fuse and li + 1 < len(layers):
            nl = layers[li + 1].layer_norm1
            x_out, hn, mn, rn = ops.gemm_nt_ln(act, s["w2"], layer.mlp.fc2.bias, x_mid, nl.weight, nl.bias, a.eps, save_stats=save)
            nxt_ln = (hn, mn, rn)
        elif lean:
            x_out, _, _ = ops.gemm_nt(act, s["w2"], layer.mlp.fc2.bias, resid=x_mid)
        else:
            _, x_out, _ = ops.gemm_nt(act, s["w2"], layer.mlp.fc2.bias, resid=x_mid, out_bf16=None, out_f32=True)
        if save:
            saved.append((x, m1, r1, h, qkv, o, lse, x_mid, m2, r2, h2, z, act))
        x = x_out
    return x, saved


# The gradient stream of the towers — the backward's counterpart of the residual stream — travels in bf16 (round 5): every
# consumer but the next LayerNorm backward is an MFMA operand anyway, and a LayerNorm backward that reads and writes ONE bf16
# stream moves 10 bytes per element instead of 16 (it is HBM-bound: 82 -> ~55 us per launch at M = 32 800, 25 launches per step).
# The stream is re-rounded at each of the 2 x layers residual joins (relative 2^-9 each, unbiased): ~0.5 % after 24 joins, below
# the 1-2 % the bf16 operands already cost the gradients (tests/test_gpu_parity2.py).  LC2IS_GRAD_STREAM=f32 keeps the fp32 chain.
_GRAD_STREAM_BF16 = __import__('os').environ.get("LC2IS_GRAD_STREAM", "bf16") != "f32"


def _stack_bwd(g32, g16, stack: _Stack, sh, saved, a: ClipArch, B: int, S: int, kbias, causal: bool, on_layer_done=None,
               final_f32: bool = True):
    """Backward of _stack_fwd.  g32/g16: gradient wrt the stack output (fp32 + bf16 twin; with the bf16 gradient stream only
    g16 is read).  Returns the gradient wrt the stack input as (fp32, bf16); the fp32 form is None when the stream is bf16 and
    `final_f32` is False."""
    C, H = a.hidden, a.heads
    D = C // H
    scale = D ** -0.5
    # weight gradients are deferred (base.WgradBatch) and leave as grouped launches, LC2IS_WGRAD_PAIR layers at a time.
    # Without a gradient reducer waiting on the layers (single GPU) the whole tower's weight gradients leave as ONE grid at the
    # end: 72 problems / 1296 tiles run as full-length blocks with ~97 % of the CUs busy, against 84 % for 216 blocks per layer.
    # Under data parallelism three layers share a launch (324 tiles: 1 full round + a finely split tail round, planner cost
    # 242 steps per layer against 287 for one layer and 225 for the tower) and report to the reducer together: its buckets
    # are >= 48 MB (two layers) anyway, so the all-reduce of a group still starts under the backward of the next group.
    if on_layer_done is not None:
        default_pair = min(3, len(stack.layers))
    else:
        default_pair = len(stack.layers)
    pair = int(__import__('os').environ.get('LC2IS_WGRAD_PAIR', str(default_pair)))
    batch, waiting = WgradBatch(), []
    batch.__enter__()
    lean = _GRAD_STREAM_BF16
    if lean:
        g32 = None
    first = stack.layers[0]
    try:
        for layer, s, sv in zip(reversed(stack.layers), reversed(sh), reversed(saved)):
            x, m1, r1, h, qkv, o, lse, x_mid, m2, r2, h2, z, act = sv
            at, mlp = layer.self_attn, layer.mlp
            # x_out = x_mid + fc2(quick_gelu(fc1(LN2(x_mid))))
            linear_bwd_params(g16, act, mlp.fc2.weight, mlp.fc2.bias)
            dz, _, _ = ops.gemm_nt(g16, s["w2T"], None, act=ops.ACT_DQUICK_GELU, aux_in=z)
            linear_bwd_params(dz, h2, mlp.fc1.weight, mlp.fc1.bias)
            dh2, _, _ = ops.gemm_nt(dz, s["w1T"], None)
            dg, accg = vec_grad(layer.layer_norm2.weight)
            db, _ = vec_grad(layer.layer_norm2.bias)
            gm32, gm16, _, _ = ops.layernorm_bwd(dh2, x_mid, layer.layer_norm2.weight, m2, r2, dres=g16 if lean else g32,
                                                 dgamma=dg, dbeta=db, accumulate=accg, want_f32=not lean,
                                                 need_param_grads=dg is not None)
            # x_mid = x + out_proj(attn(qkv(LN1(x))))
            linear_bwd_params(gm16, o, at.out_proj.weight, at.out_proj.bias)
            do, _, _ = ops.gemm_nt(gm16, s["woT"], None)
            dqkv = torch.empty_like(qkv)
            ops.attention_bwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, do, lse, B, H, S, S, D, scale, causal=causal,
                              kbias=kbias, dq=dqkv[:, :C], dk=dqkv[:, C:2 * C], dv=dqkv[:, 2 * C:])
            for j, proj in enumerate((at.q_proj, at.k_proj, at.v_proj)):
                linear_bwd_params(dqkv[:, j * C:(j + 1) * C], h, proj.weight, proj.bias)
            dh, _, _ = ops.gemm_nt(dqkv, s["wqkvT"], None)
            dg, accg = vec_grad(layer.layer_norm1.weight)
            db, _ = vec_grad(layer.layer_norm1.bias)
            g32, g16, _, _ = ops.layernorm_bwd(dh, x, layer.layer_norm1.weight, m1, r1, dres=gm16 if lean else gm32,
                                               dgamma=dg, dbeta=db, accumulate=accg,
                                               want_f32=(not lean) or (final_f32 and layer is first),
                                               need_param_grads=dg is not None)
            waiting.append(layer)
            if len(waiting) >= pair:
                batch.flush()
                if on_layer_done is not None:  # every gradient of these layers is queued: the DP reducer may start on them
                    for done in waiting:
                        on_layer_done(done)
                waiting = []
    finally:
        batch.__exit__(None, None, None)     # flushes what is left
    if on_layer_done is not None:
        for done in waiting:
            on_layer_done(done)
    return g32, g16


def _remap_legacy_keys(state_dict, prefix, inner: str):
    """Accept pre-5.x transformers checkpoints: `enc.vision_model.X` / `enc.text_model.X` -> `enc.X`."""
    legacy = prefix + "enc." + inner + "."
    for k in [k for k in state_dict if k.startswith(legacy)]:
        state_dict[prefix + "enc." + k[len(legacy):]] = state_dict.pop(k)
    state_dict.pop(prefix + "enc.embeddings.position_ids", None)


def _interp_matrix_bicubic(n_in: int, n_out: int) -> torch.Tensor:
    """U [n_out, n_in] of F.interpolate(mode="bicubic", size=n_out, align_corners=False): A = -0.75, border-clamped
    taps (host-side, init-time only)."""
    A = -0.75
    c1 = lambda x: ((A + 2) * x - (A + 3)) * x * x + 1  # noqa: E731
    c2 = lambda x: ((A * x - 5 * A) * x + 8 * A) * x - 4 * A  # noqa: E731
    U = torch.zeros(n_out, n_in, dtype=torch.float64)
    scale = n_in / n_out
    for o in range(n_out):
        src = scale * (o + 0.5) - 0.5
        i0 = math.floor(src)
        t = src - i0
        for k, wgt in enumerate((c2(t + 1), c1(t), c1(1 - t), c2(2 - t))):
            U[o, min(max(i0 - 1 + k, 0), n_in - 1)] += wgt
    return U


VIT_PRETRAINED_SIZE = 224  # model/encoder.py:9


# ---- vision -------------------------------------------------------------------------------------------
class _VisionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pixel_values, anchor, mod, keep_cls, save):
        out, saved = mod._fwd(pixel_values, keep_cls, save)
        ctx.mod, ctx.saved, ctx.keep_cls = mod, saved, keep_cls
        return out

    @staticmethod
    def backward(ctx, gout):
        ctx.mod._bwd(gout.contiguous(), ctx.saved, ctx.keep_cls)
        ctx.saved = None
        return None, None, None, None, None


class ImageEncoderCLIP(HipModule):
    """Drop-in for model/encoder.py:11-47 (ViT-B/16 for patch_size 16, ViT-L/14 for patch_size 14)."""

    keep_cls = False

    def __init__(self, in_size: int, patch_size: int = 16, arch: ClipArch | None = None) -> None:
        super().__init__()
        self.in_size, self.patch_size = in_size, patch_size
        if arch is None:
            if patch_size == 16:
                arch = VIT_B16
            elif patch_size == 14:
                arch = VIT_L14
            else:
                raise ValueError("ImageEncoderCLIP: only patch_size 16 (ViT-B/16) and 14 (ViT-L/14) have defaults")
        self.arch = arch
        self.enc = _VisionParams(arch, in_size, patch_size)
        _clip_init(self.enc, arch)

    def hidden_size(self) -> int:
        return self.arch.hidden

    def pos_emebedding_interpolate(self, tgt_size: int, ignore_index: int = 1, weight: torch.Tensor | None = None) -> torch.Tensor:
        """model/encoder.py:32-44 [sic]: 2-D bicubic resize of a position table trained at 224x224 (14x14 grid for
        patch 16) to ``tgt_size`` x ``tgt_size``, first ``ignore_index`` rows (CLS) kept.  Init-time host code."""
        w = (self.enc.embeddings.position_embedding.weight if weight is None else weight).detach().double().cpu()
        old = VIT_PRETRAINED_SIZE // self.patch_size
        keep, grid = w[:ignore_index], w[ignore_index:].reshape(old, old, -1)
        U = _interp_matrix_bicubic(old, tgt_size)
        new = torch.einsum("yi,ijc,xj->yxc", U, grid, U).reshape(tgt_size * tgt_size, -1)
        return torch.cat([keep, new], dim=0).float()

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        _remap_legacy_keys(state_dict, prefix, "vision_model")
        # a pretrained 224x224 checkpoint: resize its position table like the reference ctor does (encoder.py:24-27)
        key = prefix + "enc.embeddings.position_embedding.weight"
        old_rows = (VIT_PRETRAINED_SIZE // self.patch_size) ** 2 + 1
        mine = self.enc.embeddings.position_embedding.weight.shape[0]
        if key in state_dict and state_dict[key].shape[0] == old_rows and mine != old_rows:
            state_dict[key] = self.pos_emebedding_interpolate(self.in_size // self.patch_size, weight=state_dict[key])
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _build_shadows(self, device):
        a = self.arch
        per, entries = _stack_shadows(self.enc.encoder, a, device)
        k = 3 * self.patch_size ** 2
        kpad = (k + 63) // 64 * 64
        wp = torch.zeros(a.hidden, kpad, dtype=torch.bfloat16, device=device)
        entries.append((self.enc.embeddings.patch_embedding.weight.view(a.hidden, k), wp[:, :k], None))
        return dict(layers=per, wpatch=wp, kpad=kpad), entries

    def _fwd(self, pixel_values, keep_cls, save):
        require_cuda(pixel_values, "pixel_values")
        a, sh = self.arch, self._ensure_ready()
        B = pixel_values.shape[0]
        G = self.in_size // self.patch_size
        P = G * G
        if pixel_values.shape[-1] != self.in_size or pixel_values.shape[-2] != self.in_size:
            raise ValueError(f"Input image size ({pixel_values.shape[-2]}*{pixel_values.shape[-1]}) doesn't match "
                             f"model ({self.in_size}*{self.in_size}).")  # hf:modeling_clip.py:204-207
        emb = self.enc.embeddings
        cols = ops.patchify(pixel_values.float().contiguous(), self.patch_size, sh["kpad"])
        _, pe, _ = ops.gemm_nt(cols, sh["wpatch"], None, out_bf16=None, out_f32=True)
        x0 = ops.vit_embed_fwd(pe, emb.class_embedding, emb.position_embedding.weight, B, P)
        xb, xf, m0, r0 = ops.layernorm_fwd(x0, self.enc.pre_layrnorm.weight, self.enc.pre_layrnorm.bias, a.eps,
                                           save_stats=save, out_bf16=True if _RESID_BF16 else None,
                                           out_f32=None if _RESID_BF16 else True)
        x, saved = _stack_fwd(xb if _RESID_BF16 else xf, self.enc.encoder, sh["layers"], a, B, P + 1, None, False, save)
        if keep_cls and x.dtype == torch.float32:
            out = x.view(B, P + 1, a.hidden)
        else:   # the module's output is fp32 like the reference's: the patch rows (all rows for ...Full) leave the stream here
            n, off = (P + 1, 0) if keep_cls else (P, 1)
            out = torch.empty(B * n, a.hidden, dtype=torch.float32, device=x.device)
            ops.rows_copy(x, P + 1, off, n, 0, B, n, dst_f32=out)
            out = out.view(B, n, a.hidden)
        return out, (dict(cols=cols, x0=x0, m0=m0, r0=r0, layers=saved, B=B, P=P) if save else None)

    def _bwd(self, gout, saved, keep_cls):
        a, sh = self.arch, self._sh
        B, P = saved["B"], saved["P"]
        C = a.hidden
        if keep_cls:
            g32 = gout.reshape(B * (P + 1), C).float()
            g16 = ops.cast_bf16(g32)
        elif _GRAD_STREAM_BF16:   # the patch rows go straight into the bf16 stream (CLS rows: zero gradient)
            g32 = None
            g16 = torch.zeros(B * (P + 1), C, dtype=torch.bfloat16, device=gout.device)
            ops.rows_copy(gout.reshape(B * P, C).float().contiguous(), P, 0, P + 1, 1, B, P, dst_bf16=g16)
        else:
            g32 = torch.zeros(B * (P + 1), C, dtype=torch.float32, device=gout.device)
            ops.rows_copy(gout.reshape(B * P, C).float().contiguous(), P, 0, P + 1, 1, B, P, dst_f32=g32)
            g16 = ops.cast_bf16(g32)
        g32, g16 = _stack_bwd(g32, g16, self.enc.encoder, sh["layers"], saved["layers"], a, B, P + 1, None, False,
                              on_layer_done=self._part_grads_ready if self._part_ready_cb is not None else None,
                              final_f32=False)
        dg, accg = vec_grad(self.enc.pre_layrnorm.weight)
        db, _ = vec_grad(self.enc.pre_layrnorm.bias)
        dx0, _, _, _ = ops.layernorm_bwd(g32 if g32 is not None else g16, saved["x0"], self.enc.pre_layrnorm.weight, saved["m0"], saved["r0"],
                                         dgamma=dg, dbeta=db, accumulate=accg, want_bf16=False,
                                         need_param_grads=dg is not None)
        emb = self.enc.embeddings
        gpos, accp = grad_buf(emb.position_embedding.weight)
        gcls, accc = grad_buf(emb.class_embedding)
        if accp != accc:
            raise RuntimeError("lc2is_amd: inconsistent gradient state of the ViT embeddings")
        dpatch = ops.vit_embed_bwd(dx0, gpos, gcls, B, P, accumulate=accp)
        gw, accw = grad_buf(emb.patch_embedding.weight)
        k = 3 * self.patch_size ** 2
        if sh["kpad"] == k:
            ops.gemm_tn(dpatch, saved["cols"], gw.view(C, k), accumulate=accw)
        else:
            tmp = ops.gemm_tn(dpatch, saved["cols"])
            gw.view(C, k).copy_(tmp[:, :k]) if not accw else gw.view(C, k).add_(tmp[:, :k])
        self._grads_ready()

    def forward(self, pixel_values: torch.Tensor) -> torch.Tensor:
        anchor = self.enc.pre_layrnorm.weight
        return _VisionFn.apply(pixel_values, anchor, self, self.keep_cls, torch.is_grad_enabled() and anchor.requires_grad)


class ImageEncoderCLIPFull(ImageEncoderCLIP):
    """Drop-in for model/encoder.py:49-85: same encoder, CLS token kept -> [B, P+1, C]."""

    keep_cls = True


# ---- text ---------------------------------------------------------------------------------------------
class _TextFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input_ids, attention_mask, anchor, mod, save):
        out, saved = mod._fwd(input_ids, attention_mask, save)
        ctx.mod, ctx.saved = mod, saved
        return out

    @staticmethod
    def backward(ctx, gout):
        # autograd replays this node on the stream its forward ran on.  When that is a side stream (the composition
        # overlaps the text tower with the vision tower) the incoming gradient was produced on another stream, and
        # the parameter gradients below are written by kernels autograd knows nothing about: keep the allocator from
        # recycling `gout` early and make the stream that called backward() wait for this one before it returns.
        here = torch.cuda.current_stream(gout.device)
        gout.record_stream(here)
        torch.autograd.Variable._execution_engine.queue_callback(
            lambda: torch.cuda.current_stream(gout.device).wait_stream(here))
        ctx.mod._bwd(gout.contiguous(), ctx.saved)
        ctx.saved = None
        return None, None, None, None, None


class TextEncoderCLIP(HipModule):
    """Drop-in for model/encoder.py:87-102: CLIP text transformer, returns last_hidden_state [B,L,C]."""

    def __init__(self, patch_size: int = 16, arch: ClipArch | None = None) -> None:
        super().__init__()
        self.patch_size = patch_size
        if arch is None:
            arch = TEXT_B if patch_size == 16 else TEXT_L
        self.arch = arch
        self.enc = _TextParams(arch)
        _clip_init(self.enc, arch)

    def hidden_size(self) -> int:
        return self.arch.hidden

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        _remap_legacy_keys(state_dict, prefix, "text_model")
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _build_shadows(self, device):
        per, entries = _stack_shadows(self.enc.encoder, self.arch, device)
        return dict(layers=per), entries

    def _fwd(self, input_ids, attention_mask, save):
        require_cuda(input_ids, "input_ids")
        a, sh = self.arch, self._ensure_ready()
        B, L = input_ids.shape
        if L > a.max_pos:
            raise ValueError(f"Sequence length must be less than max_position_embeddings (got `sequence length`: "
                             f"{L} and max_position_embeddings: {a.max_pos}")  # hf CLIPTextEmbeddings.forward
        emb = self.enc.embeddings
        ids = input_ids.contiguous()
        x = ops.text_embed_fwd(ids, emb.token_embedding.weight, emb.position_embedding.weight)
        if _RESID_BF16:
            x = ops.cast_bf16(x)
        kbias = None
        if attention_mask is not None:
            kbias = torch.zeros(B, L, dtype=torch.float32, device=ids.device)
            kbias.masked_fill_(attention_mask == 0, float("-inf"))
        x, saved = _stack_fwd(x, self.enc.encoder, sh["layers"], a, B, L, kbias, True, save)
        fl = self.enc.final_layer_norm
        _, y, mf, rf = ops.layernorm_fwd(x, fl.weight, fl.bias, a.eps, save_stats=save, out_bf16=None, out_f32=True)
        return y.view(B, L, a.hidden), (dict(ids=ids, kbias=kbias, xf=x, mf=mf, rf=rf, layers=saved, B=B, L=L)
                                        if save else None)

    def _bwd(self, gout, saved):
        a, sh = self.arch, self._sh
        B, L, C = saved["B"], saved["L"], a.hidden
        fl = self.enc.final_layer_norm
        dg, accg = vec_grad(fl.weight)
        db, _ = vec_grad(fl.bias)
        g32, g16, _, _ = ops.layernorm_bwd(gout.reshape(B * L, C).float().contiguous(), saved["xf"], fl.weight,
                                           saved["mf"], saved["rf"], dgamma=dg, dbeta=db, accumulate=accg,
                                           need_param_grads=dg is not None)
        g32, _ = _stack_bwd(g32, g16, self.enc.encoder, sh["layers"], saved["layers"], a, B, L, saved["kbias"], True,
                            on_layer_done=self._part_grads_ready if self._part_ready_cb is not None else None)
        emb = self.enc.embeddings
        if emb.token_embedding.weight.requires_grad:
            gtok, acct = grad_buf(emb.token_embedding.weight)
            gpos, accp = grad_buf(emb.position_embedding.weight)
            if not acct:
                gtok.zero_()  # the scatter uses atomics
            if not accp:
                gpos.zero_()
            ops.text_embed_bwd(saved["ids"], g32, gtok, gpos, accumulate=True)
        self._grads_ready()

    def forward(self, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        anchor = self.enc.final_layer_norm.weight
        return _TextFn.apply(input_ids, attention_mask, anchor, self, torch.is_grad_enabled() and anchor.requires_grad)


class TextEncoderCLIPPooler(TextEncoderCLIP):
    """Drop-in for model/encoder.py:104-119: pooler_output = hidden state at the first EOS position
    (hf:modeling_clip.py:572-581); the gather is plain indexing on top of the HIP last_hidden_state."""

    def forward(self, input_ids: torch.Tensor, attention_mask: torch.Tensor = None) -> torch.Tensor:
        anchor = self.enc.final_layer_norm.weight
        hidden = _TextFn.apply(input_ids, attention_mask, anchor, self, torch.is_grad_enabled() and anchor.requires_grad)
        eos = (input_ids == self.arch.eos_token_id).int().argmax(dim=-1)
        return hidden[torch.arange(hidden.shape[0], device=hidden.device), eos]
