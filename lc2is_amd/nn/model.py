"""``TextToPatch`` and the ``BaseModelWithText`` composition on MI355X.

Reference: model/text_patch.py:4-19, model/model.py:12-56 (the only caller of the hot path for BASELINE
configs 1-4) and the engine contract of engine.py:81-100 (``model(inputs)`` -> ``outputs_dict["outputs"]``).

Head algebra: bicubic x4 has taps summing to 1, so it commutes with the affine
``pixel_patch.visual`` and with the prototype matmul:
    logits = up4(dec_v) W_v^T + b_v) F_t^T  ==  up4( (dec_v W_v^T + b_v) F_t^T )
The HIP path evaluates the right-hand side: two small MFMA GEMMs at 32x32 resolution, then ONE fused kernel
that upsamples the 151 class scores, and (training) computes the cross-entropy and its gradient in the same
pass.  ``literal_order=True`` is not needed for parity (the tolerance is stated in tests/) and is not provided.
"""
from __future__ import annotations

from pathlib import Path

import torch
from torch import nn

from .. import ops
from .base import HipModule, grad_buf, linear_bwd_params, require_cuda, vec_grad
from .clip import ClipArch, ImageEncoderCLIP, TextEncoderCLIP, TextEncoderCLIPPooler
from .decoder import DecoderBlock, DecoderLayer

_DATA = Path(__file__).resolve().parent.parent / "data"
KPAD = 192  # class dimension padded to a multiple of 64 (MFMA K-step of the dgrad product)


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b through the bf16 MFMA GEMM; x fp32 [..., K]."""

    @staticmethod
    def forward(ctx, x, weight, bias, w16, w16T, save):
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1]).float().contiguous()
        x16 = ops.cast_bf16(x2)
        _, y, _ = ops.gemm_nt(x16, w16, bias, out_bf16=None, out_f32=True)
        if save:
            ctx.x16, ctx.weight, ctx.bias, ctx.w16T, ctx.lead = x16, weight, bias, w16T, lead
        return y.view(*lead, weight.shape[0])

    @staticmethod
    def backward(ctx, gy):
        g2 = gy.reshape(-1, gy.shape[-1]).float().contiguous()
        g16 = ops.cast_bf16(g2)
        linear_bwd_params(g16, ctx.x16, ctx.weight, ctx.bias)
        _, dx, _ = ops.gemm_nt(g16, ctx.w16T, None, out_bf16=None, out_f32=True)
        return dx.view(*ctx.lead, dx.shape[-1]), None, None, None, None, None


class TextToPatch(HipModule):
    """Drop-in for model/text_patch.py:4-19.  forward(img, text) -> (t_feature, v_feature) — text FIRST."""

    def __init__(self, img_in: int, text_in: int, out: int = 512) -> None:
        super().__init__()
        self.textual = nn.Linear(in_features=text_in, out_features=out)
        self.visual = nn.Linear(in_features=img_in, out_features=out)

    def _build_shadows(self, device):
        bf = dict(dtype=torch.bfloat16, device=device)
        t, v = self.textual.weight, self.visual.weight
        s = dict(wt=torch.empty(*t.shape, **bf), wtT=torch.empty(t.shape[1], t.shape[0], **bf),
                 wv=torch.empty(*v.shape, **bf), wvT=torch.empty(v.shape[1], v.shape[0], **bf))
        return s, [(t, s["wt"], s["wtT"]), (v, s["wv"], s["wvT"])]

    def forward(self, img: torch.Tensor, text: torch.Tensor):
        require_cuda(img, "img")
        sh = self._ensure_ready()
        save = torch.is_grad_enabled()
        t_feature = _LinearFn.apply(text, self.textual.weight, self.textual.bias, sh["wt"], sh["wtT"], save)
        v_feature = _LinearFn.apply(img, self.visual.weight, self.visual.bias, sh["wv"], sh["wvT"], save)
        return t_feature, v_feature


# ----------------------------------------------------------------------------------------------------------
class _HeadFn(torch.autograd.Function):
    """dec_v [B,P,C] (+ class prototypes) -> NCHW logits [B,K,4g,4g]  or, with labels, the mean CE loss."""

    @staticmethod
    def forward(ctx, dec, protos, model, labels, save, ignore_index):
        out, saved = model._head_fwd(dec, labels, save, ignore_index)
        ctx.model, ctx.saved = model, saved
        return out

    @staticmethod
    def backward(ctx, gout):
        ddec = ctx.model._head_bwd(gout, ctx.saved)
        ctx.saved = None
        return ddec, None, None, None, None, None


class BaseModelWithText(HipModule):
    """Drop-in for model/model.py:12-56.

    ``forward(inputs)`` returns ``dict(outputs=logits [B,K,out,out])`` — the dict contract ``Engine`` indexes
    (engine.py:82,94,150); ``forward_tuple(inputs)`` returns the legacy
    ``(feature_t, feature_v, logits)`` of model/model.py:56; ``forward_loss(inputs, labels)`` is the fused
    training head (mean cross-entropy, identical to ``CrossEntropyLoss()(forward(inputs)["outputs"], labels)``).
    """

    def __init__(self, patch_size: int = 16, in_size: int = 224, out_size: int = 224, dropout: float = 0,
                 num_layers: int = 1, *, prototypes: torch.Tensor | str | None = None,
                 vision_arch: ClipArch | None = None, text_arch: ClipArch | None = None, nhead: int = 8,
                 dim_feedforward: int = 2048, out_dim: int = 512) -> None:
        super().__init__()
        self.patch_size, self.in_size, self.out_size = patch_size, in_size, out_size
        if out_size != 4 * (in_size // patch_size):
            raise ValueError("BaseModelWithText: out_size must be 4 * (in_size // patch_size) "
                             "(model/model.py:41-44 upsamples the patch grid by 4)")
        self.vision_encoder = ImageEncoderCLIP(in_size=in_size, patch_size=patch_size, arch=vision_arch)
        self.text_encoder = TextEncoderCLIP(patch_size=patch_size, arch=text_arch)
        if prototypes is None:
            prototypes = _DATA / "ade20k_prototypes.pt"  # model/model.py:22 (cwd-relative in the reference)
        if not isinstance(prototypes, torch.Tensor):
            prototypes = torch.load(prototypes, weights_only=True)
        self.class_prototypes = nn.Parameter(prototypes.detach().clone().float(), requires_grad=True)
        cv, ct = self.vision_encoder.hidden_size(), self.text_encoder.hidden_size()
        layer = DecoderLayer(d_model=cv, d_kv=ct, nhead=nhead, dim_feedforward=dim_feedforward, dropout=dropout,
                             batch_first=True, norm_first=True)
        self.vision_decoder = DecoderBlock(decoder_layer=layer, num_layers=num_layers)
        self.pixel_patch = TextToPatch(out=out_dim, img_in=cv, text_in=self.class_prototypes.shape[1])
        if self.class_prototypes.shape[0] > KPAD:
            raise ValueError(f"BaseModelWithText: at most {KPAD} classes are supported by the fused head")
        self.overlap_text = True     # text tower on a side stream (set False to serialise, e.g. under graph capture)
        self._text_stream = None

    # -- shadows owned by the composition: padded bf16 prototypes -------------------------------------------
    def _params_for_version(self):
        return [self.class_prototypes]

    def _build_shadows(self, device):
        K, Ct = self.class_prototypes.shape
        p16 = torch.zeros(KPAD, Ct, dtype=torch.bfloat16, device=device)
        return dict(p16=p16, K=K), [(self.class_prototypes, p16[:K], None)]

    # -- head ------------------------------------------------------------------------------------------------
    def _head_scores(self, dec16):
        sh = self._ensure_ready()
        pp = self.pixel_patch
        psh = pp._ensure_ready()
        ft16, _, _ = ops.gemm_nt(sh["p16"], psh["wt"], pp.textual.bias)            # [KPAD, out]
        fv16, _, _ = ops.gemm_nt(dec16, psh["wv"], pp.visual.bias)                  # [B*P, out]
        _, scores, _ = ops.gemm_nt(fv16, ft16, None, out_bf16=None, out_f32=True)   # [B*P, KPAD]
        return ft16, fv16, scores

    def _head_fwd(self, dec, labels, save, ignore_index):
        B, P, C = dec.shape
        g = self.in_size // self.patch_size
        K = self.class_prototypes.shape[0]
        dec16 = ops.cast_bf16(dec.reshape(B * P, C).float().contiguous())
        ft16, fv16, scores = self._head_scores(dec16)
        if labels is None:
            _, _, hi = ops.head_upsample_ce(scores, None, B, g, g, K, 4, ops.INTERP_BICUBIC, want_scores=True,
                                            want_loss=False)
            return hi, (dict(dec16=dec16, ft16=ft16, fv16=fv16, dims=(B, P, C, g, K), fused=None) if save else None)
        # the kernel writes the gradient of the SUM of the per-pixel losses and counts the pixels it kept (labels that are
        # negative, == ignore_index or >= K are skipped, head.hip); the 1/count of nn.CrossEntropyLoss's mean is folded
        # into the upstream-gradient multiply of _head_bwd, from the device-side count (no host sync, no label pass)
        loss2, dlo, _ = ops.head_upsample_ce(scores, labels.contiguous(), B, g, g, K, 4, ops.INTERP_BICUBIC,
                                             want_grad=save, ignore_index=ignore_index, grad_scale=1.0)
        loss = loss2[0] / loss2[1]
        inv_count = (1.0 / loss2[1].clamp_min(1.0)) if save else None
        return loss, (dict(dec16=dec16, ft16=ft16, fv16=fv16, dims=(B, P, C, g, K), fused=dlo, inv_count=inv_count)
                      if save else None)

    def _head_bwd(self, gout, saved):
        B, P, C, g, K = saved["dims"]
        pp = self.pixel_patch
        psh, sh = pp._sh, self._sh
        if saved["fused"] is not None:
            ds = saved["fused"]   # d(sum of pixel losses)/d(scores); gout: scalar upstream gradient of the mean loss
            ds16 = ops.cast_bf16(ds * (saved["inv_count"] if gout is None else gout * saved["inv_count"]))
        else:
            ds = ops.upsample_bwd_nchw(gout.float().contiguous(), B, g, g, K, 4, ops.INTERP_BICUBIC, KPAD)
            ds16 = ops.cast_bf16(ds)
        ft16, fv16, dec16 = saved["ft16"], saved["fv16"], saved["dec16"]
        ftT = ops.transpose_bf16(ft16)                                   # [out, KPAD]
        dfv, _, _ = ops.gemm_nt(ds16, ftT, None)                         # [B*P, out]
        dft = ops.gemm_tn(ds16, fv16)                                    # [KPAD, out] fp32
        linear_bwd_params(dfv, dec16, pp.visual.weight, pp.visual.bias)
        _, ddec, _ = ops.gemm_nt(dfv, psh["wvT"], None, out_bf16=None, out_f32=True)
        dft16 = ops.cast_bf16(dft)
        linear_bwd_params(dft16, sh["p16"], pp.textual.weight, pp.textual.bias)
        if self.class_prototypes.requires_grad:
            _, dp, _ = ops.gemm_nt(dft16, psh["wtT"], None, out_bf16=None, out_f32=True)   # [KPAD, Ct]
            gp, acc = grad_buf(self.class_prototypes)
            if acc:
                gp.add_(dp[:K])
            else:
                gp.copy_(dp[:K])
        pp._grads_ready()
        self._grads_ready()
        return ddec.view(B, P, C)

    # -- composition -----------------------------------------------------------------------------------------
    def _decode(self, inputs):
        vision_inputs = {k: v for k, v in inputs.items() if k in ["pixel_values"]}
        text_inputs = {k: v for k, v in inputs.items() if k in ["input_ids", "attention_mask"]}
        if self.overlap_text and text_inputs["input_ids"].is_cuda:
            # The text tower is ~400 tiny launches (B*L = 512 tokens at config 2) that leave most of the 256 CUs idle:
            # it runs on a side HIP stream under the vision tower's large kernels (autograd replays its backward on
            # the same stream, again beside the vision backward).
            dev = text_inputs["input_ids"].device
            main = torch.cuda.current_stream(dev)
            if self._text_stream is None or self._text_stream.device != dev:
                # HIGH priority (round 5): HIP multiplexes streams onto a few hardware queues, and two streams on one queue run
                # in order.  Which queue a pool stream lands on depends on what else created streams before it — with a
                # process group on `nccl` the side stream of this class shared the main stream's queue and the tower ran
                # SERIALISED (34.0 instead of 30.7 ms per step, profiles/r05_dp_one_gpu.txt).  Priority levels have queues of
                # their own, so a high-priority stream never shares one with the (normal-priority) stream the vision tower is on;
                # its launches are tiny and few CUs wide.  LC2IS_TEXT_STREAM_PRIO=0: a normal-priority pool stream as before.
                prio = int(__import__("os").environ.get("LC2IS_TEXT_STREAM_PRIO", "-1"))
                self._text_stream = torch.cuda.Stream(dev, priority=prio)
            side = self._text_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for t in text_inputs.values():
                    t.record_stream(side)
                enc_t = self.text_encoder(**text_inputs)                                      # model.py:32
            enc_v = self.vision_encoder(**vision_inputs)                                      # model.py:35
            main.wait_stream(side)
            enc_t.record_stream(main)
        else:
            enc_t = self.text_encoder(**text_inputs)                                          # model.py:32
            enc_v = self.vision_encoder(**vision_inputs)                                      # model.py:35
        kpm = text_inputs["attention_mask"] != 1    # model.py:38 `torch.where(mask == 1, False, True)`: the same bool tensor in one launch instead of four
        return self.vision_decoder(tgt=enc_v, memory=enc_t, memory_key_padding_mask=kpm)

    def forward(self, inputs: dict) -> dict:
        dec_v = self._decode(inputs)
        save = torch.is_grad_enabled() and dec_v.requires_grad
        logits = _HeadFn.apply(dec_v, self.class_prototypes, self, None, save, -100)
        return dict(outputs=logits)

    def forward_loss(self, inputs: dict, labels: torch.Tensor, ignore_index: int = -100) -> torch.Tensor:
        """Mean cross-entropy of the model output against ``labels`` [B,out,out] — CE(engine.py:94) fused with
        the head; never materialises the fp32 logits."""
        dec_v = self._decode(inputs)
        save = torch.is_grad_enabled() and dec_v.requires_grad
        return _HeadFn.apply(dec_v, self.class_prototypes, self, labels, save, ignore_index)

    @torch.no_grad()
    def forward_tuple(self, inputs: dict):
        """Legacy return of model/model.py:56: (feature_t [K,out], feature_v [B,out_size²,out], logits)."""
        dec_v = self._decode(inputs)
        B, P, C = dec_v.shape
        g = self.in_size // self.patch_size
        K = self.class_prototypes.shape[0]
        dec16 = ops.cast_bf16(dec_v.reshape(B * P, C).float().contiguous())
        ft16, fv16, scores = self._head_scores(dec16)
        _, _, logits = ops.head_upsample_ce(scores, None, B, g, g, K, 4, ops.INTERP_BICUBIC, want_scores=True,
                                            want_loss=False)
        fv = fv16.float()
        outs = []
        for c0 in range(0, fv.shape[1], KPAD):  # upsample the visual features 192 channels at a time
            c1 = min(c0 + KPAD, fv.shape[1])
            ld = (c1 - c0 + 63) // 64 * 64
            chunk = torch.zeros(B * P, ld, dtype=torch.float32, device=fv.device)
            chunk[:, :c1 - c0] = fv[:, c0:c1]
            _, _, hi = ops.head_upsample_ce(chunk, None, B, g, g, c1 - c0, 4, ops.INTERP_BICUBIC, want_scores=True,
                                            want_loss=False)
            outs.append(hi)
        feature_v = torch.cat(outs, dim=1).flatten(2).transpose(1, 2).contiguous()
        return ft16[:K].float(), feature_v, logits


class _ContrastiveHeadFn(torch.autograd.Function):
    """(enc_v [B,P,Cv], enc_t [Nt,Ct]) -> logits [B, (4g)^2, Nt] through the commuted head (scores at the patch grid, then
    ONE bicubic x4 of the Nt class channels)."""

    @staticmethod
    def forward(ctx, enc_v, enc_t, model, save):
        logits, saved = model._head_fwd(enc_v, enc_t, save)
        ctx.model, ctx.saved = model, saved
        return logits

    @staticmethod
    def backward(ctx, gout):
        dv, dt = ctx.model._head_bwd(gout, ctx.saved)
        ctx.saved = None
        return dv, dt, None, None


class ContrastiveModel(HipModule):
    """Drop-in for model/model.py:58-103: CLIP vision tower, POOLED CLIP text tower (one embedding per prompt), TextToPatch,
    ``logits[b, pixel, prompt] = feature_v @ feature_t^T`` on the bicubically x4-upsampled patch grid.

    ``forward(inputs) -> (feature_t, feature_v, logits)`` like the reference; ``feature_v`` ([B, out^2, out] — the tensor the
    commuted head exists to avoid) is only materialised when ``return_features`` is True (default False -> ``None``).
    Trains with ``lc2is_amd.nn.ContrastiveLoss`` (model/loss.py:39-64, which hard-codes 151 prompts)."""

    def __init__(self, patch_size: int = 16, in_size: int = 224, out_size: int = 224, dropout: float = 0, num_layers: int = 1,
                 *, vision_arch: ClipArch | None = None, text_arch: ClipArch | None = None, out_dim: int = 512) -> None:
        super().__init__()
        self.patch_size, self.in_size, self.out_size = patch_size, in_size, out_size
        if out_size != 4 * (in_size // patch_size):
            raise ValueError("ContrastiveModel: out_size must be 4 * (in_size // patch_size) (model/model.py:83-86)")
        self.vision_encoder = ImageEncoderCLIP(in_size=in_size, patch_size=patch_size, arch=vision_arch)
        self.text_encoder = TextEncoderCLIPPooler(patch_size=patch_size, arch=text_arch)
        self.pixel_patch = TextToPatch(out=out_dim, img_in=self.vision_encoder.hidden_size(),
                                       text_in=self.text_encoder.hidden_size())
        self.return_features = False

    def _params_for_version(self):
        return []

    def _head_fwd(self, enc_v, enc_t, save):
        B, P, C = enc_v.shape
        g = self.in_size // self.patch_size
        Nt, Ct = enc_t.shape
        if Nt > KPAD:
            raise ValueError(f"ContrastiveModel: at most {KPAD} prompts are supported by the fused head")
        pp = self.pixel_patch
        psh = pp._ensure_ready()
        dec16 = ops.cast_bf16(enc_v.reshape(B * P, C).float().contiguous())
        t16 = torch.zeros(KPAD, Ct, dtype=torch.bfloat16, device=enc_v.device)
        ops.cast_bf16(enc_t.float().contiguous(), t16[:Nt])
        ft16, _, _ = ops.gemm_nt(t16, psh["wt"], pp.textual.bias)                   # [KPAD, out]
        fv16, _, _ = ops.gemm_nt(dec16, psh["wv"], pp.visual.bias)                  # [B*P, out]
        _, scores, _ = ops.gemm_nt(fv16, ft16, None, out_bf16=None, out_f32=True)   # [B*P, KPAD]
        _, _, hi = ops.head_upsample_ce(scores, None, B, g, g, Nt, 4, ops.INTERP_BICUBIC, want_scores=True, want_loss=False)
        logits = hi.flatten(2).transpose(1, 2)                                       # [B, (4g)^2, Nt] (view of NCHW scores)
        saved = dict(dec16=dec16, t16=t16, ft16=ft16, fv16=fv16, dims=(B, P, C, g, Nt)) if save else None
        return logits, saved

    def _head_bwd(self, gout, saved):
        B, P, C, g, Nt = saved["dims"]
        pp = self.pixel_patch
        psh = pp._sh
        gn = gout.transpose(1, 2).reshape(B, Nt, 4 * g, 4 * g).float().contiguous()
        ds16 = ops.cast_bf16(ops.upsample_bwd_nchw(gn, B, g, g, Nt, 4, ops.INTERP_BICUBIC, KPAD))
        ft16, fv16, dec16, t16 = saved["ft16"], saved["fv16"], saved["dec16"], saved["t16"]
        dfv, _, _ = ops.gemm_nt(ds16, ops.transpose_bf16(ft16), None)               # [B*P, out]
        dft16 = ops.cast_bf16(ops.gemm_tn(ds16, fv16))                               # [KPAD, out]
        linear_bwd_params(dfv, dec16, pp.visual.weight, pp.visual.bias)
        _, dv, _ = ops.gemm_nt(dfv, psh["wvT"], None, out_bf16=None, out_f32=True)
        linear_bwd_params(dft16[:Nt], t16[:Nt], pp.textual.weight, pp.textual.bias)
        _, dt, _ = ops.gemm_nt(dft16, psh["wtT"], None, out_bf16=None, out_f32=True)   # [KPAD, Ct]
        pp._grads_ready()
        return dv.view(B, P, C), dt[:Nt].contiguous()

    def forward(self, inputs: dict):
        vision_inputs = {k: v for k, v in inputs.items() if k in ["pixel_values"]}
        text_inputs = {k: v for k, v in inputs.items() if k in ["input_ids", "attention_mask"]}
        enc_t = self.text_encoder(**text_inputs)                                                # model.py:77
        enc_v = self.vision_encoder(**vision_inputs)                                            # model.py:80
        require_cuda(enc_v, "pixel_values")
        save = torch.is_grad_enabled() and (enc_v.requires_grad or enc_t.requires_grad)
        logits = _ContrastiveHeadFn.apply(enc_v, enc_t, self, save)
        feature_t = feature_v = None
        if self.return_features:
            with torch.no_grad():
                feature_t, feature_v = self._features(enc_v, enc_t)
        return feature_t, feature_v, logits

    def _features(self, enc_v, enc_t):
        B, P, C = enc_v.shape
        g = self.in_size // self.patch_size
        pp = self.pixel_patch
        psh = pp._ensure_ready()
        ft16, _, _ = ops.gemm_nt(ops.cast_bf16(enc_t.float().contiguous()), psh["wt"], pp.textual.bias)
        fv16, _, _ = ops.gemm_nt(ops.cast_bf16(enc_v.reshape(B * P, C).float().contiguous()), psh["wv"], pp.visual.bias)
        fv = fv16.float()
        outs = []
        for c0 in range(0, fv.shape[1], KPAD):   # upsample the visual features 192 channels at a time
            c1 = min(c0 + KPAD, fv.shape[1])
            ld = (c1 - c0 + 63) // 64 * 64
            chunk = torch.zeros(B * P, ld, dtype=torch.float32, device=fv.device)
            chunk[:, :c1 - c0] = fv[:, c0:c1]
            _, _, hi = ops.head_upsample_ce(chunk, None, B, g, g, c1 - c0, 4, ops.INTERP_BICUBIC, want_scores=True, want_loss=False)
            outs.append(hi)
        return ft16.float(), torch.cat(outs, dim=1).flatten(2).transpose(1, 2).contiguous()
