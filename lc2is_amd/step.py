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

import os

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
        live = arena.finalize_grads()
        gscale = 1.0
        if self.reducer is not None:
            self.reducer.finish_step()
            gscale = 1.0 / self.reducer.world_size
        self.t += 1
        # One fused launch over the whole arena; a zero gradient leaves a parameter untouched unless weight decay is on —
        # then, like torch.optim (which skips parameters whose grad is None), only the live segments are updated so that
        # frozen / unreached parameters stay bit-identical.
        segs = [(0, arena.numel)] if (self.weight_decay == 0.0 or live == [(0, arena.numel)]) else live
        for lo, hi in segs:
            sl = slice(lo, hi)
            if self.kind == "sgd":
                ops.sgd_step(arena.flat[sl], arena.grad[sl], None if self.mom is None else self.mom[sl], self.lr,
                             self.momentum, self.weight_decay, gscale)
            else:
                ops.adamw_step(arena.flat[sl], arena.grad[sl], self.m[sl], self.v[sl], self.lr, self.betas[0],
                               self.betas[1], self.eps, self.weight_decay, self.t, gscale)
        for m in self._hip_modules:
            m.invalidate_shadows()
        return loss.detach()

    # -- HIP graph replay ------------------------------------------------------------------------------------
    def capture(self, inputs: dict, labels: torch.Tensor, warmup: int = 2):
        """Capture one full step (shadow refresh -> forward -> CE -> backward -> optimizer) into a hipGraph over
        static copies of the batch; returns ``replay(inputs, labels) -> loss`` which copies the new batch into the
        static buffers and launches the graph (one host call per step instead of ~800 kernel launches).  ``warmup`` REAL
        steps on ``inputs`` run before the capture (lazy initialisation must not happen inside it); the captured step itself
        is only recorded.
        Single-process only (the RCCL reduction is not captured)."""
        if self.reducer is not None:
            raise RuntimeError("TrainStep.capture: graph capture is only wired for single-GPU steps")
        if self.kind == "adamw":   # (checked BEFORE anything runs or is captured)
            raise RuntimeError("TrainStep.capture: AdamW bias correction is step-dependent; capture supports SGD")
        if hasattr(self.model, "overlap_text") and os.environ.get("LC2IS_GRAPH_OVERLAP", "1") == "0":
            self.model.overlap_text = False   # LC2IS_GRAPH_OVERLAP=0: one captured stream (default: the text-tower fork / join is captured too)
        from .nn.base import DropoutRng
        DropoutRng.last.clear()
        static_in = {k: v.clone() for k, v in inputs.items()}
        static_lb = labels.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):   # warm-up on the capture stream: lazy inits, workspaces, attributes
            for _ in range(warmup):
                self.step(static_in, static_lb)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if DropoutRng.last:   # dropout seeds are kernel ARGUMENTS drawn on the host: a replay would repeat one step's masks
            raise RuntimeError("TrainStep.capture: the model has active dropout / drop-path sites "
                               f"({len(DropoutRng.last)}); their per-step seeds cannot be captured — train eagerly or set the rates to 0")
        prev = getattr(self, "_captured", None)
        if prev is not None:          # capturing again: the previous captured step of THIS TrainStep is released first
            prev.release()
        graph = torch.cuda.CUDAGraph()
        t_before = self.t
        mark0 = ops.captured_tables_mark()
        with torch.cuda.graph(graph, stream=side):
            static_loss = self.step(static_in, static_lb)
        mark1 = ops.captured_tables_mark()   # the pinned descriptor-table images registered in between belong to this graph
        self.t = t_before   # capture records the step; nothing ran

        def replay(new_inputs: dict, new_labels: torch.Tensor) -> torch.Tensor:
            for k, v in new_inputs.items():
                if v is not static_in[k]:
                    static_in[k].copy_(v, non_blocking=True)
            if new_labels is not static_lb:
                static_lb.copy_(new_labels, non_blocking=True)
            graph.replay()
            self.t += 1
            return static_loss

        state = {"live": True}

        def release() -> None:
            """Destroy the captured graph and the pinned descriptor tables ITS grouped launches own (other captured steps of the
            process keep theirs).  The replay function must not be called afterwards; releasing twice is a no-op."""
            if not state["live"]:
                return
            state["live"] = False
            torch.cuda.synchronize()
            graph.reset()
            ops.release_captured_tables_range(mark0, mark1)
            if getattr(self, "_captured", None) is replay:
                self._captured = None

        replay.static_inputs, replay.static_labels, replay.graph, replay.release = static_in, static_lb, graph, release
        self._captured = replay
        return replay
