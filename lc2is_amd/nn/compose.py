"""The reference's prompt-conditioned compositions as drop-in modules (the callers of the config-5 hot path).

* ``PromptFTN``  — model/model.py:174-214: Swin backbone -> frozen pooled CLIP text embeddings refined by a ``PromptDecoder``
  over the LAST Swin stage (``text + 1e-4 * context``) -> ``FTNDecoder`` pyramid -> L2-normalise both sides, per-image class
  scores, bilinear x4.  Returns ``(None, score_map [B,K,512,512])`` like the reference (the 128 x 128 token grid is hard-coded
  there: 512 x 512 inputs).
* ``DenseClip``  — model/model.py:106-171: ``ImageEncoderCLIPFull`` (CLS row kept) + frozen pooled text tower, ``TextToPatch``,
  prompt decoder over ALL visual tokens (``text + 1e-5 * context``), normalised score map on the patch grid (no upsample) and a
  pre-norm ``DecoderBlock`` of the visual tokens over the refined text.  Returns ``(None, score_map, out)``.
  The reference builds its prompt layer as ``PromptLayer(d_model=512, nhead=8)`` — without the required ``d_kv`` and with
  ``batch_first`` left False — which cannot be constructed / run; the only reading consistent with its ``forward`` (memory =
  the 512-wide ``feature_v``, batch-first tensors) is ``d_kv = 512, batch_first=True``, which is what this module and the
  golden fixture (tools/make_golden.py) use.

Every sub-module is one of the existing HIP drop-ins (attribute names = the reference's, so its ``state_dict`` loads with
``strict=True``); the glue between them is what the reference's ``forward`` does, on device tensors.
"""
from __future__ import annotations

import torch
from torch import nn

from .base import assign_rng_names, require_cuda
from .clip import ClipArch, ImageEncoderCLIPFull, TextEncoderCLIPPooler
from .decoder import DecoderBlock, DecoderLayer, PromptDecoder, PromptLayer
from .hier import FTNDecoder
from .model import TextToPatch
from .score import KPAD, ScoreMapTail, _scores_bwd, _scores_lo
from .swin import SWIN_B, SwinArch, SwinTransformer


class PromptFTN(nn.Module):
    """Drop-in for model/model.py:174-214.  ``PromptFTN()`` = the reference's hard-coded construction (Swin-base widths
    128..1024, 8 prompt layers of d_model 512 over d_kv 1024, FTNDecoder(dim 512)); the keyword arguments are extensions for
    other backbones / depths (the reference's own ``SwinTransformer()`` default is Swin-small, whose 768-wide last stage does not
    fit the 1024 it hard-codes here — model/ftn.py:35 shows the Swin-base intent)."""

    def __init__(self, *, swin_arch: SwinArch | None = None, text_arch: ClipArch | None = None, prompt_layers: int = 8,
                 dim: int = 512, nhead: int = 8, dropout: float = 0.1) -> None:
        super().__init__()
        a = swin_arch or SWIN_B
        widths = [a.embed_dim << i for i in range(4)]
        self.textual_encoder = TextEncoderCLIPPooler(patch_size=16, arch=text_arch)   # frozen (model/model.py:178-180)
        for param in self.textual_encoder.parameters():
            param.requires_grad = False
        self.visual_encoder = SwinTransformer(a)
        self.prompt_decoder = PromptDecoder(PromptLayer(d_model=dim, d_kv=widths[-1], nhead=nhead, dropout=dropout, batch_first=True),
                                            num_layers=prompt_layers)
        self.decoder = FTNDecoder(in_dims=widths, dim=dim, dropout=dropout)
        self.tail = ScoreMapTail(4)
        assign_rng_names(self)   # prompt_decoder.* and decoder.attention_stage_*.* log their dropout sites under distinct names

    def _embeddings(self, inputs: dict):
        B = inputs["pixel_values"].shape[0]
        require_cuda(inputs["pixel_values"], "pixel_values")
        text_embeddings = self.textual_encoder(input_ids=inputs["input_ids"], attention_mask=inputs.get("attention_mask"))
        text_embeddings = text_embeddings.unsqueeze(0).expand(B, -1, -1)                          # model.py:192
        stages = self.visual_encoder(pixel_values=inputs["pixel_values"])[:4]
        if stages[0].shape[1] != 128 * 128:
            raise ValueError("PromptFTN: the reference hard-codes a 128 x 128 token grid (model/model.py:204): 512 x 512 inputs")
        visual_context = self.prompt_decoder(tgt=text_embeddings, memory=stages[-1])            # model.py:197
        text_embeddings = torch.add(text_embeddings, visual_context, alpha=1e-4)                  # model.py:199
        visual_embeddings = self.decoder(visual=list(stages), textual=text_embeddings)            # model.py:202
        return visual_embeddings, text_embeddings

    def forward(self, inputs: dict):
        visual_embeddings, text_embeddings = self._embeddings(inputs)
        return None, self.tail(visual_embeddings, text_embeddings)                                # model.py:204-214

    def forward_loss(self, inputs: dict, labels: torch.Tensor, ignore_index: int = -100) -> torch.Tensor:
        """``nn.CrossEntropyLoss()(forward(inputs)[1], labels)`` (engine.py:94) without materialising the [B,K,512,512] map."""
        visual_embeddings, text_embeddings = self._embeddings(inputs)
        return self.tail.loss(visual_embeddings, text_embeddings, labels, ignore_index)


class _ScoreGridFn(torch.autograd.Function):
    """L2-normalise visual tokens and text rows, per-image class scores on the token grid: [B,P,C] x [B,K,C] -> [B,K,h,w]."""

    @staticmethod
    def forward(ctx, visual, text, save):
        B, P, _ = visual.shape
        K = text.shape[1]
        h = int(round(P ** 0.5))
        scores, sv = _scores_lo(visual, text)                                                    # [B*P, KPAD] fp32
        ctx.sv, ctx.dims = (sv if save else None), (B, P, K, h)
        return scores.view(B, P, KPAD)[:, :, :K].transpose(1, 2).reshape(B, K, h, h)

    @staticmethod
    def backward(ctx, gout):
        B, P, K, h = ctx.dims
        ds = torch.zeros(B * P, KPAD, dtype=torch.float32, device=gout.device)
        ds.view(B, P, KPAD)[:, :, :K] = gout.reshape(B, K, P).transpose(1, 2)
        dv, dt = _scores_bwd(ds, ctx.sv)
        ctx.sv = None
        return dv, dt, None


class DenseClip(nn.Module):
    """Drop-in for model/model.py:106-171 (see the module docstring for the prompt layer's ``d_kv`` / ``batch_first``)."""

    def __init__(self, patch_size: int, in_size: int, out_size: int, *, vision_arch: ClipArch | None = None,
                 text_arch: ClipArch | None = None, num_layers: int = 8, dim: int = 512, nhead: int = 8,
                 prompt_dropout: float = 0.1, dim_feedforward: int = 2048) -> None:
        super().__init__()
        self.patch_size, self.in_size, self.out_size = patch_size, in_size, out_size
        self.vision_encoder = ImageEncoderCLIPFull(in_size=in_size, patch_size=patch_size, arch=vision_arch)
        self.text_encoder = TextEncoderCLIPPooler(patch_size=patch_size, arch=text_arch)          # frozen (model.py:115-117)
        for param in self.text_encoder.parameters():
            param.requires_grad = False
        hv = self.vision_encoder.hidden_size()
        self.text_patch = TextToPatch(out=dim, img_in=hv, text_in=self.text_encoder.hidden_size())
        self.prompt_decoder = PromptDecoder(PromptLayer(d_model=dim, d_kv=dim, nhead=nhead, dim_feedforward=dim_feedforward,
                                                        dropout=prompt_dropout, batch_first=True), num_layers=num_layers)
        self.vision_decoder = DecoderBlock(decoder_layer=DecoderLayer(d_model=hv, d_kv=dim, nhead=nhead, dim_feedforward=dim_feedforward,
                                                                      batch_first=True, norm_first=True), num_layers=num_layers)
        assign_rng_names(self)   # prompt_decoder.layers.i.* / vision_decoder.layers.i.*: distinct dropout-site names

    def forward(self, inputs: dict):
        B = inputs["pixel_values"].shape[0]
        require_cuda(inputs["pixel_values"], "pixel_values")
        enc_t = self.text_encoder(input_ids=inputs["input_ids"], attention_mask=inputs.get("attention_mask"))   # model.py:132
        enc_v = self.vision_encoder(pixel_values=inputs["pixel_values"])                                          # model.py:135
        feature_t, feature_v = self.text_patch(img=enc_v, text=enc_t)                                             # model.py:144
        feature_t = feature_t.unsqueeze(0).expand(B, -1, -1)
        v_context = self.prompt_decoder(tgt=feature_t, memory=feature_v)                                          # model.py:148
        text_embeddings = torch.add(feature_t, v_context, alpha=1e-5)                                             # model.py:151
        visual_embeddings = feature_v[:, 1:, :]                                                                   # model.py:155 (CLS row dropped)
        save = torch.is_grad_enabled() and (visual_embeddings.requires_grad or text_embeddings.requires_grad)
        score_map = _ScoreGridFn.apply(visual_embeddings, text_embeddings, save)                                  # model.py:161-163
        out = self.vision_decoder(tgt=enc_v, memory=text_embeddings)                                              # model.py:166
        return None, score_map, out
