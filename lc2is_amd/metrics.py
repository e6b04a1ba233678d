"""Device-side counterpart of the reference's label-size mIoU (metrics.py:82-102).

``compute_mIOU(outputs [N,K,h,w] logits, labels [N,h,w])``: per image, bicubic x4 of the logits (HIP upsample kernel),
argmax (Softmax2d is monotone, so it is skipped), nearest x4 of the labels, per-class intersection / union counts in one
HIP pass, IoU averaged over the classes present in the label except ``ignore_index``, then over images.
Returns ``dict(mIOU_label=float)`` like the reference.  (torchmetrics is not needed.)"""
from __future__ import annotations

import torch

from . import ops


def per_image_mIOU(outputs: torch.Tensor, labels: torch.Tensor, n_cls: int = 151, ignore_index: int | None = 0) -> torch.Tensor:
    """The per-image values whose mean ``compute_mIOU`` returns (float64 [N] on the device; NaN for an image whose label holds
    nothing but ``ignore_index``).  ``Evaluator`` accumulates these batch by batch instead of keeping every batch's logits."""
    if not outputs.is_cuda:
        raise RuntimeError("lc2is_amd.metrics: outputs must be on the GPU (no CPU path)")
    N, K, h, w = outputs.shape
    ld = (K + 63) // 64 * 64
    lo = torch.zeros(N * h * w, ld, dtype=torch.float32, device=outputs.device)
    lo[:, :K] = outputs.float().permute(0, 2, 3, 1).reshape(N * h * w, K)
    _, _, hi = ops.head_upsample_ce(lo, None, N, h, w, K, 4, ops.INTERP_BICUBIC, want_scores=True, want_loss=False)
    counts = ops.miou_counts(hi, labels, 4).to(torch.float64)           # [N, 3, K]
    inter, pred, lab = counts[:, 0], counts[:, 1], counts[:, 2]
    union = pred + lab - inter
    iou = torch.where(union > 0, inter / union.clamp_min(1), torch.zeros_like(union))
    present = lab > 0
    if ignore_index is not None:
        present[:, ignore_index] = False
    # an image whose label holds nothing but ignore_index: the reference takes the mean of an EMPTY selection (metrics.py:94-97),
    # which is NaN, and the mean over images (:101) inherits it — 0 / 0 here reproduces that instead of scoring the image 0
    return (iou * present).sum(1) / present.sum(1).to(torch.float64)


def compute_mIOU(outputs: torch.Tensor, labels: torch.Tensor, n_cls: int = 151, ignore_index: int | None = 0) -> dict:
    return dict(mIOU_label=float(per_image_mIOU(outputs, labels, n_cls, ignore_index).mean().item()))
