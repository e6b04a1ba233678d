"""CPU oracle for the LC2IS hot path — TEST INFRASTRUCTURE ONLY.

A plain fp32 (or fp64) PyTorch restatement of what the reference computes on the path named by
BASELINE.json:north_star.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product (``lc2is_amd``) never does.

Pinning: the reference's own tests hold no golden vectors, so this restatement is pinned
against outputs of the reference itself, produced in the build container by ``tools/make_golden.py``
(which imports the reference's modules) and committed as tensors under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks every function here against them.

Every function is written from explicit tensor algebra (matmul / softmax / mean / var / gather) — no
``transformers`` import, no ``nn.Transformer*`` — and cites the reference lines it follows.  "hf:" =
transformers/models/clip/modeling_clip.py (5.15.0), "torch:" = torch/nn (2.10).

State dicts use the reference's own parameter names.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------------------------------------
def layer_norm(x: Tensor, w: Tensor, b: Tensor | None, eps: float = 1e-5) -> Tensor:
    """torch:nn/functional.py layer_norm — biased variance over the last dim."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    y = (x - mu) / torch.sqrt(var + eps) * w
    return y + b if b is not None else y


def linear(x: Tensor, w: Tensor, b: Tensor | None = None) -> Tensor:
    y = x @ w.transpose(-1, -2)
    return y + b if b is not None else y


def quick_gelu(x: Tensor) -> Tensor:
    """hf:activations.py:122-123  x * sigmoid(1.702 x)."""
    return x * torch.sigmoid(1.702 * x)


def mha_core(q: Tensor, k: Tensor, v: Tensor, nhead: int, scale: float, bias: Tensor | None,
             pdrop: Tensor | None = None) -> Tensor:
    """softmax(scale * Q K^T + bias) V per head.  q [B,Sq,C], k/v [B,Sk,C]; bias broadcastable to [B,H,Sq,Sk].
    Follows hf:modeling_clip.py:259-277 (eager_attention_forward) and the attention core of
    torch:nn/functional.py multi_head_attention_forward.  pdrop [B,H,Sq,Sk] (optional): the dropout multiplier
    keep / (1 - p) applied to the attention probabilities (training mode, F.dropout after the softmax)."""
    B, Sq, C = q.shape
    Sk = k.shape[1]
    d = C // nhead
    qh = q.reshape(B, Sq, nhead, d).transpose(1, 2)
    kh = k.reshape(B, Sk, nhead, d).transpose(1, 2)
    vh = v.reshape(B, Sk, nhead, d).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * scale
    if bias is not None:
        s = s + bias
    p = torch.softmax(s, dim=-1)
    if pdrop is not None:
        p = p * pdrop
    return (p @ vh).transpose(1, 2).reshape(B, Sq, C)


def _cubic_coeffs(t: Tensor, A: float = -0.75):
    """torch upsample_bicubic2d cubic convolution coefficients (ATen UpSample.h get_cubic_upsample_coefficients)."""
    def c1(x):
        return ((A + 2) * x - (A + 3)) * x * x + 1

    def c2(x):
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A

    return c2(t + 1), c1(t), c1(1 - t), c2(2 - t)


def interp_matrix(n_in: int, n_out: int, mode: str, scale: float | None = None, dtype=torch.float32) -> Tensor:
    """U [n_out, n_in] with out = U @ in, for F.interpolate(align_corners=False) along one axis.
    bicubic: A=-0.75, taps clamped to the border; bilinear: source index clamped at 0.
    `scale` = 1/scale_factor when the caller passed scale_factor (model/model.py:43), else n_in/n_out (size=)."""
    s = (n_in / n_out) if scale is None else scale
    U = torch.zeros(n_out, n_in, dtype=torch.float64)
    for o in range(n_out):
        src = s * (o + 0.5) - 0.5
        if mode == "bicubic":
            i0 = math.floor(src)
            t = src - i0
            ws = _cubic_coeffs(torch.tensor(t, dtype=torch.float64))
            for k in range(4):
                idx = min(max(i0 - 1 + k, 0), n_in - 1)
                U[o, idx] += float(ws[k])
        elif mode == "bilinear":
            src = max(src, 0.0)
            i0 = int(src)
            i1 = min(i0 + 1, n_in - 1)
            l1 = src - i0
            U[o, i0] += 1 - l1
            U[o, i1] += l1
        else:
            raise ValueError(mode)
    return U.to(dtype)


def upsample2d(x: Tensor, scale_factor: int | None = None, size: int | None = None, mode: str = "bicubic") -> Tensor:
    """x [B,C,h,w] -> [B,C,H,W]; separable, exactly F.interpolate(mode, align_corners=False)."""
    h, w = x.shape[-2:]
    if scale_factor is not None:
        H, W, sc = h * scale_factor, w * scale_factor, 1.0 / scale_factor
    else:
        H = W = size
        sc = None
    Uy = interp_matrix(h, H, mode, sc, x.dtype)
    Ux = interp_matrix(w, W, mode, sc if scale_factor is not None else None, x.dtype)
    return Uy @ x @ Ux.transpose(0, 1)


def cross_entropy(logits: Tensor, labels: Tensor, ignore_index: int = -100) -> Tensor:
    """nn.CrossEntropyLoss() on [B,C,*] logits / [B,*] labels, mean over counted positions
    (evaluate.py:68, engine.py:82,94,150)."""
    C = logits.shape[1]
    lg = logits.movedim(1, -1).reshape(-1, C)
    lb = labels.reshape(-1)
    lse = torch.logsumexp(lg, dim=-1)
    keep = lb != ignore_index
    picked = lg.gather(1, lb.clamp(0, C - 1).unsqueeze(1)).squeeze(1)
    return ((lse - picked) * keep).sum() / keep.sum()


# ----------------------------------------------------------------------------------------------------------
# CLIP encoders (model/encoder.py -> hf CLIPVisionModel / CLIPTextModel)
# ----------------------------------------------------------------------------------------------------------
@dataclass
class ClipCfg:
    hidden: int
    heads: int
    layers: int
    eps: float = 1e-5
    patch: int = 16          # vision only
    eos_token_id: int = 49407  # text only


def _clip_layer(sd: dict, pre: str, x: Tensor, cfg: ClipCfg, bias: Tensor | None) -> Tensor:
    """hf:modeling_clip.py:362-383 CLIPEncoderLayer.forward (pre-LN), attention :298-335, MLP :346-350."""
    h = layer_norm(x, sd[pre + "layer_norm1.weight"], sd[pre + "layer_norm1.bias"], cfg.eps)
    q = linear(h, sd[pre + "self_attn.q_proj.weight"], sd[pre + "self_attn.q_proj.bias"])
    k = linear(h, sd[pre + "self_attn.k_proj.weight"], sd[pre + "self_attn.k_proj.bias"])
    v = linear(h, sd[pre + "self_attn.v_proj.weight"], sd[pre + "self_attn.v_proj.bias"])
    a = mha_core(q, k, v, cfg.heads, (cfg.hidden // cfg.heads) ** -0.5, bias)
    x = x + linear(a, sd[pre + "self_attn.out_proj.weight"], sd[pre + "self_attn.out_proj.bias"])
    h = layer_norm(x, sd[pre + "layer_norm2.weight"], sd[pre + "layer_norm2.bias"], cfg.eps)
    h = quick_gelu(linear(h, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"]))
    return x + linear(h, sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])


def clip_vision_tokens(sd: dict, pre: str, pixel_values: Tensor, cfg: ClipCfg) -> Tensor:
    """hf CLIPVisionModel.forward last_hidden_state (hf:613-656): conv patchify (no bias) + CLS + learned
    positions (:202-218) -> pre_layrnorm -> layers; NO post_layernorm on tokens.  Returns [B, P+1, C]."""
    w = sd[pre + "embeddings.patch_embedding.weight"]  # [C,3,p,p]
    B, _, H, W = pixel_values.shape
    p = cfg.patch
    G = H // p
    x = pixel_values[:, :, :G * p, :G * p].reshape(B, 3, G, p, G, p).permute(0, 2, 4, 1, 3, 5).reshape(B, G * G, 3 * p * p)
    patches = x @ w.reshape(w.shape[0], -1).transpose(0, 1)
    cls = sd[pre + "embeddings.class_embedding"].expand(B, 1, -1)
    x = torch.cat([cls, patches], dim=1) + sd[pre + "embeddings.position_embedding.weight"]
    x = layer_norm(x, sd[pre + "pre_layrnorm.weight"], sd[pre + "pre_layrnorm.bias"], cfg.eps)
    for i in range(cfg.layers):
        x = _clip_layer(sd, f"{pre}encoder.layers.{i}.", x, cfg, None)
    return x


def image_encoder_clip(sd: dict, pre: str, pixel_values: Tensor, cfg: ClipCfg) -> Tensor:
    """ImageEncoderCLIP.forward: drop the CLS token (model/encoder.py:29-30)."""
    return clip_vision_tokens(sd, pre + "enc.", pixel_values, cfg)[:, 1:, :]


def image_encoder_clip_full(sd: dict, pre: str, pixel_values: Tensor, cfg: ClipCfg) -> Tensor:
    """ImageEncoderCLIPFull.forward keeps CLS (model/encoder.py:67-68)."""
    return clip_vision_tokens(sd, pre + "enc.", pixel_values, cfg)


def pos_embedding_interpolate(pos: Tensor, tgt_size: int, patch: int = 16, pretrained: int = 224) -> Tensor:
    """ImageEncoderCLIP.pos_emebedding_interpolate (model/encoder.py:32-44): bicubic resize (size=) of the
    [1+g*g, C] table's grid part, CLS row kept."""
    g = pretrained // patch
    grid = pos[1:].reshape(g, g, -1).permute(2, 0, 1).unsqueeze(0)
    new = upsample2d(grid, size=tgt_size, mode="bicubic")
    new = new.squeeze(0).permute(1, 2, 0).reshape(tgt_size * tgt_size, -1)
    return torch.cat([pos[:1], new], dim=0)


def clip_text(sd: dict, pre: str, input_ids: Tensor, attention_mask: Tensor | None, cfg: ClipCfg):
    """hf CLIPTextModel.forward (hf:513-586): token+position embedding, causal ∧ padding additive mask,
    pre-LN layers, final_layer_norm; pooled = state at the first EOS position (:572-581)."""
    B, L = input_ids.shape
    x = sd[pre + "embeddings.token_embedding.weight"][input_ids] + sd[pre + "embeddings.position_embedding.weight"][:L]
    neg = torch.finfo(x.dtype).min
    bias = torch.zeros(B, 1, L, L, dtype=x.dtype)
    bias = bias.masked_fill(torch.ones(L, L, dtype=torch.bool).triu(1), neg)
    if attention_mask is not None:
        bias = bias.masked_fill((attention_mask == 0)[:, None, None, :], neg)
    for i in range(cfg.layers):
        x = _clip_layer(sd, f"{pre}encoder.layers.{i}.", x, cfg, bias)
    x = layer_norm(x, sd[pre + "final_layer_norm.weight"], sd[pre + "final_layer_norm.bias"], cfg.eps)
    eos = (input_ids == cfg.eos_token_id).int().argmax(dim=-1)
    return x, x[torch.arange(B), eos]


def text_encoder_clip(sd, pre, input_ids, attention_mask, cfg) -> Tensor:
    """TextEncoderCLIP.forward (model/encoder.py:98-99)."""
    return clip_text(sd, pre + "enc.", input_ids, attention_mask, cfg)[0]


def text_encoder_clip_pooler(sd, pre, input_ids, attention_mask, cfg) -> Tensor:
    """TextEncoderCLIPPooler.forward (model/encoder.py:115-116)."""
    return clip_text(sd, pre + "enc.", input_ids, attention_mask, cfg)[1]


# ----------------------------------------------------------------------------------------------------------
# decoder (model/decoder.py:9-21 -> torch TransformerDecoderLayer / MultiheadAttention)
# ----------------------------------------------------------------------------------------------------------
def decoder_layer(sd: dict, pre: str, tgt: Tensor, memory: Tensor, nhead: int, norm_first: bool = True,
                  memory_key_padding_mask: Tensor | None = None, eps: float = 1e-5, drop: dict | None = None) -> Tensor:
    """DecoderLayer.forward = torch:nn/modules/transformer.py:1131-1145 with multihead_attn rebuilt for
    kdim=vdim=d_kv (separate q/k/v projection weights, packed in_proj_bias).  Biases are optional: under
    torch 2.10 the reference creates self_attn / linear1-2 / norm1-3 without them (SURVEY.md §2 drift #1).
    drop (training mode, :1158-1199): multipliers keep / (1 - p) for the six dropout sites — "sa_p", "ca_p"
    [B,H,Sq,Sk] on the attention probabilities, "d1", "d2", "d3" [B,Sq,C] on the branch outputs, "ff" [B,Sq,F] between
    the activation and linear2; a missing key means no dropout at that site.  "relu_mask": see ff() below."""
    g = lambda k: sd.get(pre + k)  # noqa: E731
    dm = (lambda name, t: t * drop[name] if (drop is not None and name in drop) else t)  # noqa: E731
    C = tgt.shape[-1]
    scale = (C // nhead) ** -0.5

    def sa(x):
        w, b = g("self_attn.in_proj_weight"), g("self_attn.in_proj_bias")
        qkv = linear(x, w, b)
        q, k, v = qkv.split(C, dim=-1)
        a = mha_core(q, k, v, nhead, scale, None, drop.get("sa_p") if drop else None)
        return dm("d1", linear(a, g("self_attn.out_proj.weight"), g("self_attn.out_proj.bias")))

    def ca(x):
        b = g("multihead_attn.in_proj_bias")
        bq, bk, bv = (b.split(C) if b is not None else (None, None, None))
        q = linear(x, g("multihead_attn.q_proj_weight"), bq)
        k = linear(memory, g("multihead_attn.k_proj_weight"), bk)
        v = linear(memory, g("multihead_attn.v_proj_weight"), bv)
        bias = None
        if memory_key_padding_mask is not None:
            bias = torch.zeros(memory.shape[0], 1, 1, memory.shape[1], dtype=x.dtype)
            bias = bias.masked_fill(memory_key_padding_mask[:, None, None, :], float("-inf"))
        a = mha_core(q, k, v, nhead, scale, bias, drop.get("ca_p") if drop else None)
        return dm("d2", linear(a, g("multihead_attn.out_proj.weight"), g("multihead_attn.out_proj.bias")))

    def ff(x):
        z = linear(x, g("linear1.weight"), g("linear1.bias"))
        # "relu_mask" [B,Sq,F] (diagnostic input of the parity tests): the 0/1 activation pattern of ANOTHER run replaces this
        # run's own sign test, act = z * mask — separates "which units fired" (a discontinuous choice that bf16 inputs flip
        # for units next to zero) from the arithmetic downstream of it
        act = z * drop["relu_mask"] if (drop is not None and "relu_mask" in drop) else torch.relu(z)
        return dm("d3", linear(dm("ff", act), g("linear2.weight"), g("linear2.bias")))

    n = lambda i, x: layer_norm(x, g(f"norm{i}.weight"), g(f"norm{i}.bias"), eps)  # noqa: E731
    x = tgt
    if norm_first:
        x = x + sa(n(1, x))
        x = x + ca(n(2, x))
        x = x + ff(n(3, x))
    else:
        x = n(1, x + sa(x))
        x = n(2, x + ca(x))
        x = n(3, x + ff(x))
    return x


def decoder_block(sd, pre, tgt, memory, nhead, num_layers, norm_first=True, memory_key_padding_mask=None, drops=None):
    """DecoderBlock.forward (model/decoder.py:20-21): num_layers layers, no final norm.  drops: per-layer dropout multipliers."""
    x = tgt
    for i in range(num_layers):
        x = decoder_layer(sd, f"{pre}layers.{i}.", x, memory, nhead, norm_first, memory_key_padding_mask,
                          drop=drops[i] if drops else None)
    return x


def text_to_patch(sd, pre, img: Tensor, text: Tensor):
    """TextToPatch.forward (model/text_patch.py:14-19) — returns (t_feature, v_feature), text first."""
    return (linear(text, sd[pre + "textual.weight"], sd[pre + "textual.bias"]),
            linear(img, sd[pre + "visual.weight"], sd[pre + "visual.bias"]))


# ----------------------------------------------------------------------------------------------------------
# composition (model/model.py:27-56) and the training step (engine.py:69-123)
# ----------------------------------------------------------------------------------------------------------
@dataclass
class BaseCfg:
    in_size: int
    out_size: int
    patch: int
    vision: ClipCfg
    text: ClipCfg
    dec_heads: int = 8
    dec_layers: int = 1


def base_model_with_text(sd: dict, inputs: dict, cfg: BaseCfg, dec_drops: list | None = None):
    """BaseModelWithText.forward, literal operation order (model/model.py:27-56).
    Returns (feature_t [K,512], feature_v [B,out²,512], logits [B,K,out,out]).
    dec_drops: per-decoder-layer multiplier dicts of decoder_layer (dropout masks / the diagnostic "relu_mask")."""
    enc_t = text_encoder_clip(sd, "text_encoder.", inputs["input_ids"], inputs["attention_mask"], cfg.text)
    enc_v = image_encoder_clip(sd, "vision_encoder.", inputs["pixel_values"], cfg.vision)
    kpm = inputs["attention_mask"] != 1
    dec_v = decoder_block(sd, "vision_decoder.", enc_v, enc_t, cfg.dec_heads, cfg.dec_layers, True, kpm, drops=dec_drops)
    B, P, C = dec_v.shape
    H = cfg.in_size // cfg.patch
    x = dec_v.reshape(B, H, H, C).permute(0, 3, 1, 2)
    x = upsample2d(x, scale_factor=4, mode="bicubic")
    x = x.permute(0, 2, 3, 1).reshape(B, cfg.out_size * cfg.out_size, C)
    feature_t, feature_v = text_to_patch(sd, "pixel_patch.", x, sd["class_prototypes"])
    mm = feature_v @ feature_t.transpose(1, 0)
    logits = mm.reshape(B, cfg.out_size, cfg.out_size, -1).permute(0, 3, 1, 2)
    return feature_t, feature_v, logits


def contrastive_model(sd: dict, inputs: dict, cfg: BaseCfg):
    """ContrastiveModel.forward (model/model.py:72-103): vision tower, POOLED text tower (one row per prompt), bicubic x4
    of the patch grid, TextToPatch, logits = feature_v @ feature_t^T.  Returns (feature_t [Nt,out], feature_v
    [B,out^2,out], logits [B,out^2,Nt])."""
    enc_t = text_encoder_clip_pooler(sd, "text_encoder.", inputs["input_ids"], inputs.get("attention_mask"), cfg.text)
    enc_v = image_encoder_clip(sd, "vision_encoder.", inputs["pixel_values"], cfg.vision)
    B, P, C = enc_v.shape
    H = cfg.in_size // cfg.patch
    x = upsample2d(enc_v.reshape(B, H, H, C).permute(0, 3, 1, 2), scale_factor=4, mode="bicubic")
    x = x.permute(0, 2, 3, 1).reshape(B, cfg.out_size * cfg.out_size, C)
    feature_t, feature_v = text_to_patch(sd, "pixel_patch.", x, enc_t)
    return feature_t, feature_v, feature_v @ feature_t.transpose(0, 1)


def train_step_sgd(sd: dict, inputs: dict, labels: Tensor, cfg: BaseCfg, lr: float):
    """One iteration of Engine.train_loop (engine.py:78-104) without aux loss: zero_grad -> forward -> CE
    (mean) -> backward -> SGD step.  Returns (loss, logits, grads dict, new params dict)."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    full = dict(sd)
    full.update(params)
    _, _, logits = base_model_with_text(full, inputs, cfg)
    loss = cross_entropy(logits, labels)
    loss.backward()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in params.items()}
    new = {k: (p.detach() - lr * grads[k]) for k, p in params.items()}
    return loss.detach(), logits.detach(), grads, new


# ----------------------------------------------------------------------------------------------------------
# losses (model/loss.py) and the parity metric (metrics.py:82-102)
# ----------------------------------------------------------------------------------------------------------
def auxiliary_loss(inp: Tensor, target: Tensor, ignore_index: int = -100) -> Tensor:
    """AuxiliaryLoss.forward (model/loss.py:17-21): bilinear resize to the label size (size=H), then CE."""
    H = target.shape[1]
    return cross_entropy(upsample2d(inp, size=H, mode="bilinear"), target, ignore_index)


def npair_loss(x: Tensor, x_pos: Tensor, x_neg: Tensor) -> Tensor:
    """NPairLoss.forward (model/loss.py:30-37) with the default mean reduction."""
    pos = x @ x_pos.transpose(0, 1)
    neg = (x @ x_neg.transpose(0, 1)).sum(-1, keepdim=True)
    return (pos / (pos + neg)).sum(-1).mean()


def contrastive_loss(outputs: Tensor, labels: Tensor, num_classes: int = 151):
    """ContrastiveLoss.forward (model/loss.py:46-64).  outputs [B,HW,K]; labels [B,H,W].
    loss_visual: CE over classes per pixel; loss_textual: CE over dim 1 (=H!) of the [B,H,W,K] view against
    one-hot float targets — the reference passes the class axis LAST, so torch softmaxes over dim 1."""
    B, HW, K = outputs.shape
    H = int(math.sqrt(HW))
    out_textual = outputs.reshape(B, H, H, K)
    out_visual = outputs.transpose(-2, -1).reshape(B, K, H, H)
    onehot = torch.nn.functional.one_hot(labels, num_classes).to(outputs.dtype)  # [B,H,W,K]
    logp = torch.log_softmax(out_textual, dim=1)
    loss_textual = -(onehot * logp).sum(dim=1).mean()
    loss_visual = cross_entropy(out_visual, labels)
    return (loss_textual + loss_visual) / 2, loss_visual, loss_textual


def compute_miou(outputs: Tensor, labels: Tensor, n_class: int = 151, ignore_index: int = 0) -> float:
    """metrics.compute_mIOU (metrics.py:82-102) without torchmetrics: per image, bicubic x4 of the logits and
    nearest x4 of the labels, argmax (Softmax2d is monotone), per-class IoU (JaccardIndex average="none"),
    mean over the classes present in the label excluding ignore_index; mean over images."""
    vals = []
    for i in range(outputs.shape[0]):
        up = upsample2d(outputs[i:i + 1], scale_factor=4, mode="bicubic")[0]
        lab = labels[i].repeat_interleave(4, 0).repeat_interleave(4, 1)
        pred = up.argmax(0)
        present = [c for c in torch.unique(lab).tolist() if c != ignore_index]
        ious = []
        for c in present:
            inter = ((pred == c) & (lab == c)).sum().item()
            union = ((pred == c) | (lab == c)).sum().item()
            ious.append(inter / union if union else 0.0)
        vals.append(sum(ious) / len(ious) if ious else float("nan"))
    return float(sum(vals) / len(vals))


def evaluate(sd: dict, batches: list, cfg: BaseCfg) -> dict:
    """Engine.evaluate / eval_loop (engine.py:125-168) for BaseModelWithText: per batch forward under no_grad and
    ``nn.CrossEntropyLoss()`` (mean over batches of the per-batch losses, :158-165), then ``compute_metrics`` on the
    concatenated outputs (:128-130) = metrics.compute_mIOU (metrics.py:82-102).  ``batches``: list of
    (inputs dict incl. "label", metas)."""
    losses, outs, labs = [], [], []
    with torch.no_grad():
        for inputs, _ in batches:
            inputs = dict(inputs)
            labels = inputs.pop("label")
            _, _, logits = base_model_with_text(sd, inputs, cfg)
            losses.append(float(cross_entropy(logits, labels)))
            outs.append(logits)
            labs.append(labels)
    outputs, labels = torch.cat(outs), torch.cat(labs)
    return dict(eval_loss=sum(losses) / len(losses), eval_mIOU_label=compute_miou(outputs, labels),
                outputs=outputs, labels=labels)


# ----------------------------------------------------------------------------------------------------------
# multi-scale decoders (BASELINE config 5): model/hierarchical.py, model/decoder.py:36-134
# ----------------------------------------------------------------------------------------------------------
def _mha_packed(sd: dict, pre: str, q_in: Tensor, kv_in: Tensor, nhead: int, bias: Tensor | None = None,
                pdrop: Tensor | None = None) -> Tensor:
    """nn.MultiheadAttention with a packed in_proj_weight [3C,C] (q from q_in, k/v from kv_in), optional biases
    (absent under the torch-2.10 `bias` drift)."""
    C = q_in.shape[-1]
    w = sd[pre + "in_proj_weight"]
    b = sd.get(pre + "in_proj_bias")
    bq, bk, bv = (b.split(C) if b is not None else (None, None, None))
    q = linear(q_in, w[:C], bq)
    k = linear(kv_in, w[C:2 * C], bk)
    v = linear(kv_in, w[2 * C:], bv)
    a = mha_core(q, k, v, nhead, (C // nhead) ** -0.5, bias, pdrop)
    return linear(a, sd[pre + "out_proj.weight"], sd.get(pre + "out_proj.bias"))


def sr_reduce(sd: dict, pre: str, x: Tensor) -> Tensor:
    """Conv2d(d, d, kernel=2, stride=2) over the token grid + LayerNorm (model/hierarchical.py:189-193,212-216)."""
    B, P, C = x.shape
    H = int(P ** 0.5)
    w = sd[pre + "sr.weight"]                                     # [C, C, 2, 2]
    xg = x.reshape(B, H // 2, 2, H // 2, 2, C).permute(0, 1, 3, 5, 2, 4).reshape(B, (H // 2) ** 2, C * 4)
    r = xg @ w.reshape(C, C * 4).transpose(0, 1) + sd[pre + "sr.bias"]
    return layer_norm(r, sd[pre + "norm.weight"], sd[pre + "norm.bias"], 1e-5)


def _dm(drop, name, t):
    """Apply the dropout multiplier keep / (1 - p) of site `name` when one is given (training mode)."""
    return t * drop[name] if (drop is not None and name in drop) else t


def _relu(drop, z):
    """relu(z) — or, when the parity tests feed "relu_mask" (the 0/1 activation pattern ANOTHER run used, see decoder_layer's
    ff()), z times that pattern: a pre-activation within rounding distance of 0 flips between a bf16 run and this fp32 one, and
    with the pattern fixed what is left of a gradient difference is arithmetic."""
    return z * drop["relu_mask"] if (drop is not None and "relu_mask" in drop) else torch.relu(z)


def sr_cross_layer(sd: dict, pre: str, tgt: Tensor, memory: Tensor, nhead: int, eps: float = 1e-5, drop: dict | None = None) -> Tensor:
    """SRTransformerCrossA / SRTransformerDecoder forward (post-norm TransformerDecoderLayer whose self-attention
    keys/values are the spatially reduced tokens): model/hierarchical.py:201-225, model/decoder.py:113-134,
    torch:nn/modules/transformer.py:1147-1156."""
    g = lambda k: sd.get(pre + k)  # noqa: E731
    pd = (lambda k: drop.get(k) if drop else None)  # noqa: E731
    x = tgt
    x = layer_norm(x + _dm(drop, "d1", _mha_packed(sd, pre + "self_attn.", x, sr_reduce(sd, pre, x), nhead, None, pd("sa_p"))),
                   g("norm1.weight"), g("norm1.bias"), eps)
    x = layer_norm(x + _dm(drop, "d2", _mha_packed(sd, pre + "multihead_attn.", x, memory, nhead, None, pd("ca_p"))),
                   g("norm2.weight"), g("norm2.bias"), eps)
    ff = linear(_dm(drop, "ff", _relu(drop, linear(x, g("linear1.weight"), g("linear1.bias")))), g("linear2.weight"),
                g("linear2.bias"))
    return layer_norm(x + _dm(drop, "d3", ff), g("norm3.weight"), g("norm3.bias"), eps)


def sr_self_layer(sd: dict, pre: str, src: Tensor, nhead: int, eps: float = 1e-5, drop: dict | None = None) -> Tensor:
    """SRTransformerSelfA forward (post-norm TransformerEncoderLayer), model/hierarchical.py:174-199."""
    g = lambda k: sd.get(pre + k)  # noqa: E731
    x = src
    x = layer_norm(x + _dm(drop, "d1", _mha_packed(sd, pre + "self_attn.", x, sr_reduce(sd, pre, x), nhead, None,
                                                   drop.get("sa_p") if drop else None)), g("norm1.weight"), g("norm1.bias"), eps)
    ff = linear(_dm(drop, "ff", _relu(drop, linear(x, g("linear1.weight"), g("linear1.bias")))), g("linear2.weight"),
                g("linear2.bias"))
    return layer_norm(x + _dm(drop, "d3", ff), g("norm2.weight"), g("norm2.bias"), eps)


def _up2_tokens(x: Tensor, factor: int = 2) -> Tensor:
    B, P, C = x.shape
    H = int(P ** 0.5)
    y = upsample2d(x.reshape(B, H, H, C).permute(0, 3, 1, 2), scale_factor=factor, mode="bilinear")
    return y.permute(0, 2, 3, 1).reshape(B, P * factor * factor, C)


def attn_block(sd: dict, pre: str, x: Tensor, memory: Tensor | None, nhead: int, depth: int, layer_key: str,
               drops: list | None = None) -> Tensor:
    """CrossABlock / SelfABlock / FTNBlock: `depth` applications of ONE shared layer, then bilinear x2
    (model/hierarchical.py:140-172, model/decoder.py:103-111).  drops: per-application dropout multipliers."""
    for it in range(depth):
        d = drops[it] if drops else None
        x = sr_cross_layer(sd, pre + layer_key, x, memory, nhead, drop=d) if memory is not None else \
            sr_self_layer(sd, pre + layer_key, x, nhead, drop=d)
    return _up2_tokens(x)


def hierarchical(sd: dict, pre: str, visual: list, textual: Tensor | None, nhead: int, depth=(1, 1, 1),
                 layer_key: str = "layers.0.", drops: dict | None = None) -> Tensor:
    """HierarchicalCrossA / HierarchicalSelfA / FTNDecoder forward (model/hierarchical.py:37-69,99-131,
    model/decoder.py:62-94): top-down pyramid, reads only visual[0] and visual[3].
    drops (diagnostic / training-mode input of the tests): {block prefix, e.g. "attention_stage_4.0.": [per-application dict]}."""
    dr = lambda k: (drops or {}).get(k)  # noqa: E731
    L = lambda name, x: linear(x, sd[pre + name + ".weight"], sd[pre + name + ".bias"])  # noqa: E731
    t4 = visual[3]
    t3 = L("linear_stage_3", _up2_tokens(t4))
    t2 = L("linear_stage_2", _up2_tokens(t3))
    t1 = L("linear2_stage_1", visual[0])
    t4, t3, t2 = L("linear2_stage_4", t4), L("linear2_stage_3", t3), L("linear2_stage_2", t2)
    for i in range(3):
        t4 = attn_block(sd, f"{pre}attention_stage_4.{i}.", t4, textual, nhead, depth[2], layer_key, dr(f"attention_stage_4.{i}."))
    for i in range(2):
        t3 = attn_block(sd, f"{pre}attention_stage_3.{i}.", t3, textual, nhead, depth[1], layer_key, dr(f"attention_stage_3.{i}."))
    t2 = attn_block(sd, f"{pre}attention_stage_2.0.", t2, textual, nhead, depth[0], layer_key, dr("attention_stage_2.0."))
    return t1 + t2 + t3 + t4


# ----------------------------------------------------------------------------------------------------------
# the older FTN pyramid: model/ftn.py:67-157
# ----------------------------------------------------------------------------------------------------------
def std_decoder_layer(sd: dict, pre: str, tgt: Tensor, memory: Tensor, nhead: int, eps: float = 1e-5,
                      drop: dict | None = None) -> Tensor:
    """nn.TransformerDecoderLayer(d, nhead, batch_first=True) forward, post-norm, relu, dropout off
    (torch:nn/modules/transformer.py:1147-1156), as instantiated at model/ftn.py:135."""
    g = lambda k: sd.get(pre + k)  # noqa: E731
    pd = (lambda k: drop.get(k) if drop else None)  # noqa: E731
    x = tgt
    x = layer_norm(x + _dm(drop, "d1", _mha_packed(sd, pre + "self_attn.", x, x, nhead, None, pd("sa_p"))), g("norm1.weight"),
                   g("norm1.bias"), eps)
    x = layer_norm(x + _dm(drop, "d2", _mha_packed(sd, pre + "multihead_attn.", x, memory, nhead, None, pd("ca_p"))),
                   g("norm2.weight"), g("norm2.bias"), eps)
    ff = linear(_dm(drop, "ff", torch.relu(linear(x, g("linear1.weight"), g("linear1.bias")))), g("linear2.weight"),
                g("linear2.bias"))
    return layer_norm(x + _dm(drop, "d3", ff), g("norm3.weight"), g("norm3.bias"), eps)


def _up2_grid(x: Tensor, h: int) -> Tensor:
    """tokens as an h x (P/h) grid -> bilinear x2 -> tokens (model/ftn.py:153-155; h is NOT updated between layers)."""
    B, P, C = x.shape
    y = upsample2d(x.reshape(B, h, P // h, C).permute(0, 3, 1, 2), scale_factor=2, mode="bilinear")
    return y.permute(0, 2, 3, 1).reshape(B, 4 * P, C)


def ftn_transformer(sd: dict, pre: str, x: Tensor, h: int, repeat: int, sr_ratio: int, upsample: bool,
                    nhead: int, drops: list | None = None) -> Tensor:
    """ftn.Transformer.forward (model/ftn.py:143-157): memory = LayerNorm(Conv2d(k=sr, s=sr)(x grid)) of the block
    input (x itself when sr_ratio == 1); `repeat` decoder layers, each followed by the x2 upsample when enabled."""
    memory = sr_reduce(sd, pre, x) if sr_ratio > 1 else x
    for r in range(repeat):
        x = std_decoder_layer(sd, f"{pre}trans.{r}.layers.0.", x, memory, nhead, drop=drops[r] if drops else None)
        if upsample:
            x = _up2_grid(x, h)
    return x


def ftn_decoder(sd: dict, pre: str, xs: list) -> Tensor:
    """ftn.Decoder.forward (model/ftn.py:103-129): grids [128,64,32,16]; stages 1 and 2 add the bilinear x2 of the
    NEXT stage's raw input; attentions[0] is never applied."""
    H = [128, 64, 32, 16]
    L = lambda name, x: linear(x, sd[pre + name + ".weight"], sd[pre + name + ".bias"])  # noqa: E731
    out = [L(f"linears.{i}", xs[i]) for i in range(4)]
    for i in (1, 2):
        out[i] = out[i] + _up2_grid(xs[i + 1], H[i + 1])
    cfg = {1: (1, 8), 2: (2, 8), 3: (3, 8)}   # stage -> (repeat, nhead), model/ftn.py:84-89
    end = L("linears2.0", out[0])
    for i in range(1, 4):
        rep, nh = cfg[i]
        end = end + ftn_transformer(sd, f"{pre}attentions.{i}.", L(f"linears2.{i}", out[i]), H[i], rep, 2, True, nh)
    return end


def score_map_tail(visual_embeddings: Tensor, text_embeddings: Tensor, scale: int = 4) -> Tensor:
    """model/final.py:350-356 (same ops at model/model.py:204-212, model/ftn.py:56-62): tokens -> NCHW, L2-normalise
    both sides over channels, einsum('bchw,bkc->bkhw'), bilinear x4."""
    B, P, C = visual_embeddings.shape
    H = int(P ** 0.5)
    v = visual_embeddings.reshape(B, H, H, C)
    v = v / v.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    t = text_embeddings / text_embeddings.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    score = torch.einsum("bhwc,bkc->bkhw", v, t)
    return upsample2d(score, scale_factor=scale, mode="bilinear")


# ----------------------------------------------------------------------------------------------------------
# Swin backbone: model/encoder.py:121-131 -> hf:models/swin/modeling_swin.py (transformers 5.15 key names)
# ----------------------------------------------------------------------------------------------------------
@dataclass
class SwinCfg:
    embed_dim: int = 96
    depths: tuple = (2, 2, 18, 2)
    num_heads: tuple = (3, 6, 12, 24)
    window: int = 7
    patch: int = 4
    eps: float = 1e-5


def swin_relative_position_index(ws: int) -> Tensor:
    """SwinRelativePositionBias._create_relative_position_index (modeling_swin.py:350-365), flattened [ws^2 * ws^2]."""
    coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1).view(-1)


def swin_block(sd: dict, pre: str, x: Tensor, H: int, W: int, nH: int, ws: int, shift: int, eps: float,
               drop_path: Tensor | None = None) -> Tensor:
    """SwinLayer.forward (modeling_swin.py:529-574).  drop_path [B] (training mode): SwinDropPath's per-sample multiplier
    keep / (1 - p) on the attention branch (:567; the MLP branch only has nn.Dropout(hidden_dropout_prob = 0), :571)."""
    B, L, C = x.shape
    if min(H, W) <= ws:
        raise NotImplementedError("oracle swin_block: grids no larger than the window are outside the path (512^2 input)")
    g = lambda k: sd[pre + k]  # noqa: E731
    h = layer_norm(x, g("layernorm_before.weight"), g("layernorm_before.bias"), eps).view(B, H, W, C)
    pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
    h = torch.nn.functional.pad(h, (0, 0, 0, pad_r, 0, pad_b))
    Hp, Wp = H + pad_b, W + pad_r
    if shift > 0:
        h = torch.roll(h, shifts=(-shift, -shift), dims=(1, 2))
    win = h.view(B, Hp // ws, ws, Wp // ws, ws, C).transpose(2, 3).reshape(-1, ws * ws, C)
    S, D = ws * ws, C // nH
    q = linear(win, g("attention.q_proj.weight"), g("attention.q_proj.bias")).view(-1, S, nH, D).transpose(1, 2)
    k = linear(win, g("attention.k_proj.weight"), g("attention.k_proj.bias")).view(-1, S, nH, D).transpose(1, 2)
    v = linear(win, g("attention.v_proj.weight"), g("attention.v_proj.bias")).view(-1, S, nH, D).transpose(1, 2)
    table = g("attention.relative_position_bias.relative_position_bias_table")          # [(2ws-1)^2, nH]
    bias = table[swin_relative_position_index(ws)].view(S, S, nH).permute(2, 0, 1)        # [nH, S, S]
    logits = (q @ k.transpose(-1, -2)) * (D ** -0.5) + bias[None]
    if shift > 0:                                                                        # get_attn_mask (:584-607)
        hr = (torch.arange(Hp) >= Hp - ws).long() + (torch.arange(Hp) >= Hp - shift).long()
        wr = (torch.arange(Wp) >= Wp - ws).long() + (torch.arange(Wp) >= Wp - shift).long()
        img = (hr[:, None] * 3 + wr[None, :]).to(x.dtype)
        mw = img.view(Hp // ws, ws, Wp // ws, ws).transpose(1, 2).reshape(-1, S)
        am = mw[:, None, :] - mw[:, :, None]
        am = torch.where(am != 0, torch.full_like(am, -100.0), torch.zeros_like(am))     # [nW, S, S]
        nW = am.shape[0]
        logits = (logits.view(B, nW, nH, S, S) + am[None, :, None]).view(-1, nH, S, S)
    o = (torch.softmax(logits, dim=-1) @ v).transpose(1, 2).reshape(-1, S, C)
    o = linear(o, g("attention.o_proj.weight"), g("attention.o_proj.bias"))
    o = o.view(B, Hp // ws, Wp // ws, ws, ws, C).transpose(2, 3).reshape(B, Hp, Wp, C)
    if shift > 0:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    o = o[:, :H, :W, :].reshape(B, H * W, C)
    if drop_path is not None:
        o = o * drop_path.view(B, 1, 1)
    x = x + o
    h2 = layer_norm(x, g("layernorm_after.weight"), g("layernorm_after.bias"), eps)
    h2 = linear(h2, g("mlp.fc1.weight"), g("mlp.fc1.bias"))
    h2 = 0.5 * h2 * (1.0 + torch.erf(h2 * 0.7071067811865476))                          # hf:activations "gelu" (exact)
    return x + linear(h2, g("mlp.fc2.weight"), g("mlp.fc2.bias"))


def swin_patch_merging(sd: dict, pre: str, x: Tensor, H: int, W: int, eps: float = 1e-5) -> Tensor:
    """SwinPatchMerging.forward (modeling_swin.py:309-326): pad to even, concat 2x2 neighbours (col-major order),
    LayerNorm(4C), Linear(4C -> 2C, no bias)."""
    B, L, C = x.shape
    f = x.view(B, H, W, C)
    if H % 2 or W % 2:
        f = torch.nn.functional.pad(f, (0, 0, 0, W % 2, 0, H % 2))
    f = torch.cat([f[:, r::2, c::2, :] for c in range(2) for r in range(2)], dim=-1).reshape(B, -1, 4 * C)
    f = layer_norm(f, sd[pre + "norm.weight"], sd[pre + "norm.bias"], eps)
    return linear(f, sd[pre + "reduction.weight"], None)


def swin_hidden_states(sd: dict, pre: str, pixel_values: Tensor, cfg: SwinCfg, n_out: int = 4, drop_paths: dict | None = None) -> list:
    """SwinTransformer.forward (model/encoder.py:129-131): SwinModel(..., output_hidden_states=True).hidden_states[:4]
    = (patch embedding output, stage 1/2/3 outputs AFTER their patch merging); `pre` is the reference's attribute path
    ("encoder.").  Stage 4 and the final layernorm never reach those four tensors and are not evaluated."""
    B, _, Hi, Wi = pixel_values.shape
    p = cfg.patch
    if Hi % p or Wi % p:
        pixel_values = torch.nn.functional.pad(pixel_values, (0, (p - Wi % p) % p, 0, (p - Hi % p) % p))
    x = torch.nn.functional.conv2d(pixel_values, sd[pre + "embeddings.patch_embeddings.projection.weight"],
                                   sd[pre + "embeddings.patch_embeddings.projection.bias"], stride=p)
    H, W = x.shape[2], x.shape[3]
    x = x.flatten(2).transpose(1, 2)
    x = layer_norm(x, sd[pre + "embeddings.norm.weight"], sd[pre + "embeddings.norm.bias"], cfg.eps)
    outs = [x]
    for si in range(n_out - 1):
        for bi in range(cfg.depths[si]):
            x = swin_block(sd, f"{pre}encoder.layers.{si}.blocks.{bi}.", x, H, W, cfg.num_heads[si], cfg.window,
                           0 if bi % 2 == 0 else cfg.window // 2, cfg.eps,
                           drop_paths.get((si, bi)) if drop_paths else None)   # {(stage, block): [B] multipliers}
        x = swin_patch_merging(sd, f"{pre}encoder.layers.{si}.downsample.", x, H, W, cfg.eps)
        H, W = (H + 1) // 2, (W + 1) // 2
        outs.append(x)
    return outs
