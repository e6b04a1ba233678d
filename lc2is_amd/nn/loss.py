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


class _ContrastiveFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, outputs, labels, H):
        B, HW, K = outputs.shape
        W = HW // H
        x = outputs.reshape(B * HW, K).float().contiguous()
        lab = labels.reshape(B * HW).contiguous()
        sums = torch.zeros(2, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        ops.rows_ce(x, lab, loss_sum=sums[0:1], dx=dx, grad_scale=0.5 / (B * HW))
        ops.cols_ce(x, lab, B, H, W, K, sums[1:2], dx=dx, grad_scale=0.5 / (B * W * K))
        loss_visual = sums[0] / (B * HW)
        loss_textual = sums[1] / (B * W * K)
        ctx.dx, ctx.shape = dx, outputs.shape
        ctx.mark_non_differentiable(loss_visual, loss_textual)
        return (loss_textual + loss_visual) / 2, loss_visual, loss_textual

    @staticmethod
    def backward(ctx, g, _gv, _gt):
        return (ctx.dx * g).view(ctx.shape), None, None


class ContrastiveLoss(nn.Module):
    """Drop-in for model/loss.py:39-64.  outputs [B,HW,K], labels [B,H,W] -> (mean, loss_visual, loss_textual);
    the textual term reproduces nn.CrossEntropyLoss on a [B,H,W,K] input with one-hot float targets (class axis =
    dim 1, i.e. a softmax over image rows), exactly what the reference computes."""

    def __init__(self, weight=None, size_average=None, ignore_index: int = -100, reduce=None, reduction: str = "mean",
                 label_smoothing: float = 0) -> None:
        super().__init__()
        if weight is not None or reduction != "mean" or label_smoothing != 0:
            raise NotImplementedError("lc2is_amd ContrastiveLoss: default CrossEntropyLoss configuration only")

    def forward(self, outputs: torch.Tensor, labels: torch.Tensor):
        require_cuda(outputs, "outputs")
        H = int(round(outputs.shape[1] ** 0.5))
        if outputs.shape[2] != 151:
            raise ValueError("ContrastiveLoss: the reference hard-codes num_classes=151 (model/loss.py:55)")
        return _ContrastiveFn.apply(outputs, labels, H)


class _NPairFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, x_pos, x_neg):
        xs = [t.float().contiguous() for t in (x, x_pos, x_neg)]
        ctx.save_for_backward(*xs)
        return ops.npair(*xs)

    @staticmethod
    def backward(ctx, dres):
        x, xp, xn = ctx.saved_tensors
        return ops.npair_bwd(x, xp, xn, dres.float().contiguous())


class NPairLoss(nn.Module):
    """Drop-in for model/loss.py:23-37 (forward and backward on the HIP path; unused by every composition)."""

    def __init__(self, reduction=torch.mean) -> None:
        super().__init__()
        self.reduction = reduction

    def forward(self, x: torch.Tensor, x_pos: torch.Tensor, x_neg: torch.Tensor):
        require_cuda(x, "x")
        res = _NPairFn.apply(x, x_pos, x_neg)
        return self.reduction(res) if self.reduction else res
