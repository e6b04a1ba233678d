"""Losses of the hot path on MI355X — drop-ins for ``nn.CrossEntropyLoss()`` as used by the reference
(evaluate.py:68, engine.py:82,94,150) and for ``model/loss.py``'s ``AuxiliaryLoss``.

Both take NCHW fp32 logits and int64 labels like the reference; reduction is 'mean' over non-ignored pixels.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import ops
from .base import require_cuda


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        lg = logits.float().contiguous()
        lb = labels.contiguous()
        loss2, lse = ops.ce_nchw_fwd(lg, lb, ignore_index)
        ctx.saved = (lg, lb, lse, loss2, ignore_index)
        return loss2[0] / loss2[1]

    @staticmethod
    def backward(ctx, g):
        lg, lb, lse, loss2, ignore_index = ctx.saved
        scale = (g / loss2[1]).reshape(1).float().contiguous()  # device scalar: upstream grad / counted pixels
        return ops.ce_nchw_bwd(lg, lb, lse, scale, 1.0, ignore_index), None, None


class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss() for [B,C,H,W] logits / [B,H,W] labels on the HIP path (mean reduction)."""

    def __init__(self, weight=None, size_average=None, ignore_index: int = -100, reduce=None, reduction: str = "mean",
                 label_smoothing: float = 0.0) -> None:
        super().__init__()
        if weight is not None or reduction != "mean" or label_smoothing != 0.0:
            raise NotImplementedError("lc2is_amd CrossEntropyLoss: only the reference's default configuration "
                                      "(no class weights, mean reduction, no label smoothing) is implemented")
        self.ignore_index = ignore_index

    def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        require_cuda(input, "logits")
        if input.dim() != 4 or target.dim() != 3:
            raise ValueError("lc2is_amd CrossEntropyLoss expects [B,C,H,W] logits and [B,H,W] labels")
        return _CEFn.apply(input, target, self.ignore_index)


class _AuxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, target, ignore_index, S):
        B, K, h, w = inp.shape
        ld = (K + 63) // 64 * 64
        lo = torch.zeros(B * h * w, ld, dtype=torch.float32, device=inp.device)
        lo[:, :K] = inp.float().permute(0, 2, 3, 1).reshape(B * h * w, K)
        n = float(B * h * S * w * S)
        loss2, dlo, _ = ops.head_upsample_ce(lo, target.contiguous(), B, h, w, K, S, ops.INTERP_BILINEAR,
                                             want_grad=True, ignore_index=ignore_index, grad_scale=1.0 / n)
        ctx.saved = (dlo, loss2, n, (B, K, h, w))
        return loss2[0] / loss2[1]

    @staticmethod
    def backward(ctx, g):
        dlo, loss2, n, (B, K, h, w) = ctx.saved
        d = dlo[:, :K].reshape(B, h, w, K).permute(0, 3, 1, 2) * (g * n / loss2[1])
        return d.contiguous(), None, None, None


class AuxiliaryLoss(CrossEntropyLoss):
    """Drop-in for model/loss.py:12-21: bilinear-resize the low-resolution score map to the label size, then
    cross-entropy — one fused HIP pass (the resized map is never materialised)."""

    def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        require_cuda(input, "input")
        B, H, W = target.shape
        h = input.shape[-1]
        if input.shape[-2] != h or H != W or H % h or (H // h) not in (4, 8, 16) or input.shape[1] > 192:
            raise NotImplementedError("lc2is_amd AuxiliaryLoss: square maps, integer scale 4/8/16, <= 192 classes")
        return _AuxFn.apply(input, target, self.ignore_index, H // h)
