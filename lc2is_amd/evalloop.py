"""The evaluation pass — this repo's counterpart of ``Engine.evaluate`` / ``Engine.eval_loop`` (reference
engine.py:125-168) plus the label-size metric it feeds (``metrics.segmentation_metrics`` -> ``compute_mIOU``,
metrics.py:45-58,82-102).

    ev = Evaluator(model, eval_loader, criterion, aux_criterion=None, compute_metrics=segmentation_metrics)
    metrics = ev.evaluate()        # {"eval_loss": ..., ["eval_aux_loss": ...], "eval_mIOU_label": ...}

Semantics kept from the reference:
  * ``model.eval()``; per batch ``inputs, metas = data``; ``labels = inputs.pop("label")``; ``torch.no_grad()`` forward;
    ``eval_loss = criterion(outputs_dict["outputs"], labels)``; the aux loss (x 0.4) when the model returns
    ``low_score_map`` (engine.py:143-156);
  * the loss metrics are the MEAN OVER BATCHES of the per-batch losses (engine.py:158-165: a list of ``.item()``s,
    ``np.array(v).mean()``), not a pixel-weighted mean;
  * ``compute_metrics(outputs=all_outputs, labels=all_labels)`` on the concatenated outputs, keys prefixed ``eval_``
    (engine.py:128-130).
Differences (MI355X): outputs and labels stay ON THE DEVICE (the reference moves every batch to the host and re-concatenates
the growing tensor each step, engine.py:162-163 — O(n^2) host copies); the per-batch losses are kept as device scalars and
read once at the end (one sync per evaluation instead of one ``.item()`` per batch, engine.py:159); the metric is the HIP
``lc2is_amd.metrics.compute_mIOU`` (fused bicubic x4 + argmax + per-class counts).
"""
from __future__ import annotations

from typing import Callable, Iterable

import torch
from torch import nn

from . import metrics as _metrics


def segmentation_metrics(outputs: torch.Tensor, labels: torch.Tensor, n_clas: int = 151, ignore_index: int | None = 0,
                         **_unused) -> dict:
    """metrics.segmentation_metrics (metrics.py:45-58), label-size branch: ``dict(mIOU_label=...)``.  The ground-truth-size
    branch (``compute_gt_mIOU``) needs the original images' label maps, which ``eval_loop`` never supplies in the reference
    either (engine.py:166; SURVEY.md §2 staleness)."""
    return _metrics.compute_mIOU(outputs=outputs, labels=labels, n_cls=n_clas, ignore_index=ignore_index)


class Evaluator:
    """``Engine``'s evaluation half with the same constructor argument names (engine.py:15-21)."""

    def __init__(self, model: nn.Module, eval_loader: Iterable, criterion: nn.Module, aux_criterion: nn.Module | None = None,
                 compute_metrics: Callable | None = segmentation_metrics, device="cuda", keep_outputs: bool = False) -> None:
        self.model = model
        self.eval_loader = eval_loader
        self.criterion = criterion
        self.aux_criterion = aux_criterion
        self.compute_metrics = compute_metrics
        self.device = torch.device(device)
        # With the default metric (a mean of per-image values) nothing but those values is kept between batches: the reference
        # concatenates every batch's logits (on the host, engine.py:162-163), which on the device would be ~20 GB (+ as much again
        # for the cat) over the 2000-image ADE20K validation split at 128 x 128.  ``keep_outputs=True`` (or a custom
        # ``compute_metrics``, whose contract is ``compute_metrics(outputs=..., labels=...)``) restores the concatenation.
        self.keep_outputs = keep_outputs or (compute_metrics is not None and compute_metrics is not segmentation_metrics)
        self.model.to(self.device)

    def evaluate(self) -> dict:
        eval_metrics, eval_outputs = self.eval_loop()
        if self.compute_metrics is not None:
            if "per_image_mIOU" in eval_outputs:
                m = dict(mIOU_label=float(eval_outputs["per_image_mIOU"].mean().item()))
            else:
                m = self.compute_metrics(**eval_outputs)
            eval_metrics = {**eval_metrics, **{"eval_" + k: v for k, v in m.items()}}
        return eval_metrics

    def eval_loop(self) -> tuple[dict, dict]:
        self.model.eval()
        losses: dict[str, list[torch.Tensor]] = {}
        outs, labs = [], []
        for data in self.eval_loader:
            inputs, _metas = data
            inputs = {k: v.to(self.device, non_blocking=True) for k, v in inputs.items()}
            labels = inputs.pop("label")
            with torch.no_grad():
                outputs_dict = self.model(inputs)
                step = dict(eval_loss=self.criterion(outputs_dict["outputs"], labels))
                if "low_score_map" in outputs_dict.keys():
                    step["eval_aux_loss"] = self.aux_criterion(outputs_dict["low_score_map"], labels) * 0.4
            for k, v in step.items():
                losses.setdefault(k, []).append(v.detach().float().reshape(()))
            if self.keep_outputs or self.compute_metrics is None:
                outs.append(outputs_dict["outputs"])
                labs.append(labels)
            else:
                outs.append(_metrics.per_image_mIOU(outputs_dict["outputs"], labels))
        eval_metrics = {k: float(torch.stack(v).mean().item()) for k, v in losses.items()}
        if self.keep_outputs or self.compute_metrics is None:
            eval_outputs = dict(outputs=torch.cat(outs), labels=torch.cat(labs))
        else:
            eval_outputs = dict(per_image_mIOU=torch.cat(outs))
        return eval_metrics, eval_outputs
