"""The training step — this repo's counterpart of ``Engine.train_loop`` (reference engine.py:69-123).

One call to :meth:`TrainStep.step` is one iteration of the reference's hot loop:
    optimizer.zero_grad()                                   engine.py:78
    outputs_dict = model(inputs)                            engine.py:93
    loss = criterion(outputs_dict["outputs"], labels)       engine.py:94     (mean CE; fused with the head here)
    loss.backward()                                         engine.py:100    (+ NEW: DP gradient all-reduce)
    optimizer.step()                                        engine.py:101
and returns the loss as a DEVICE tensor (the reference's per-step ``.item()`` sync, engine.py:108, is left to
the caller's logging cadence).

MI355X layout: all parameters and gradients live in one flat fp32 arena (``ParamArena``), so the optimizer is
a single fused HIP launch and the data-parallel reduction is over one contiguous buffer, issued per module
(head, decoder, vision, text) as soon as that module's backward has run, on RCCL's own stream, overlapping
the remaining backward (``lc2is_amd.dp.GradReducer``).
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops
from .nn.base import HipModule, ParamArena


class TrainStep:
    def __init__(self, model: nn.Module, *, optimizer: str = "sgd", lr: float = 1e-5, momentum: float = 0.0,
                 weight_decay: float = 0.0, betas=(0.9, 0.999), eps: float = 1e-8, reducer=None,
                 ignore_index: int = -100) -> None:
        self.model = model
        self.arena = ParamArena(model)
        self.kind = optimizer.lower()
        if self.kind not in ("sgd", "adamw"):
            raise ValueError("TrainStep: optimizer must be 'sgd' or 'adamw'")
        self.lr, self.momentum, self.weight_decay, self.betas, self.eps = lr, momentum, weight_decay, betas, eps
        dev = self.arena.flat.device
        self.mom = torch.zeros_like(self.arena.flat) if (self.kind == "sgd" and momentum != 0.0) else None
        if self.kind == "adamw":
            self.m, self.v = torch.zeros_like(self.arena.flat), torch.zeros_like(self.arena.flat)
        self.t = 0
        self.reducer = reducer
        self.ignore_index = ignore_index
        self._hip_modules = [m for m in model.modules() if isinstance(m, HipModule)]
        if reducer is not None:
            reducer.attach(model, self.arena)
        self._dev = dev

    def step(self, inputs: dict, labels: torch.Tensor) -> torch.Tensor:
        arena = self.arena
        arena.zero_grad(set_to_none=True)
        if self.reducer is not None:
            self.reducer.begin_step()
        loss = self.model.forward_loss(inputs, labels, self.ignore_index)
        loss.backward()
        arena.finalize_grads()
        gscale = 1.0
        if self.reducer is not None:
            self.reducer.finish_step()
            gscale = 1.0 / self.reducer.world_size
        self.t += 1
        if self.kind == "sgd":
            ops.sgd_step(arena.flat, arena.grad, self.mom, self.lr, self.momentum, self.weight_decay, gscale)
        else:
            ops.adamw_step(arena.flat, arena.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps,
                           self.weight_decay, self.t, gscale)
        for m in self._hip_modules:
            m.invalidate_shadows()
        return loss.detach()
